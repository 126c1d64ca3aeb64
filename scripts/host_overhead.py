"""Host-side cost of one batch-1 DepthModel.forward: the C library is replaced by a stub that returns success (like
tests/test_host_logic_stub.py), so what is timed and profiled is the Python launch path alone -- runs without a GPU."""
import contextlib, ctypes, sys, time, cProfile, pstats
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import _lib, synthetic, depth_model as dm
calls=[]
class Stream: cuda_stream = 0
class Fn:
    def __init__(self, name): self.name=name; self.res=_lib.SIGNATURES[name][0]
    def __call__(self, *args):
        calls.append(self.name)
        if self.name == "sr_conv_prefers_wino": return 1
        if self.res is ctypes.c_char_p: return b"stub"
        return 4096 if self.res is ctypes.c_size_t else 0
class Lib:
    def __init__(self): self.c={}
    def __getattr__(self, name):
        f=Fn(name); setattr(self,name,f); return f
L=Lib()
_lib.lib=lambda: L
_lib.require_device_f32=lambda *a, **k: None
_lib.stream_ptr=lambda dev=None: ctypes.c_void_p(0)
torch.cuda.device=lambda d: contextlib.nullcontext()
torch.cuda.current_stream=lambda d=None: Stream()
b,k,h,w=1,7,64,96
opts = dm.default_options(image_width=w, image_height=h, model_num_views=k+1, matching_num_depth_bins=64)
model = dm.DepthModel(opts).eval()
model.prior_on_side_stream=False
inp = synthetic.cost_volume_inputs(b, k, 16, h // 4, w // 4, seed=2)
eye = torch.eye(4).expand(b, 4, 4).contiguous()
cur = {"image_b3hw": torch.randn(b, 3, h, w), "invK_s1_b44": inp["cur_invK"], "cam_T_world_b44": eye, "world_T_cam_b44": eye}
src = {"image_b3hw": torch.randn(b, k, 3, h, w), "K_s1_b44": inp["src_Ks"], "cam_T_world_b44": inp["src_extrinsics"], "world_T_cam_b44": inp["src_poses"]}
def run():
    with torch.inference_mode():
        return model("test", cur, src)
for _ in range(3): run()
del calls[:]
t0=time.perf_counter(); n=10
for _ in range(n): run()
t=(time.perf_counter()-t0)/n
print(f"host time per forward {t*1e3:.2f} ms, {len(calls)/n:.0f} C calls -> {t*1e6/(len(calls)/n):.1f} us per call")
pr=cProfile.Profile(); pr.enable()
for _ in range(5): run()
pr.disable()
st=pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
