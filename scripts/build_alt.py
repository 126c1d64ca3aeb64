"""Builds an ALTERNATIVE libsimplerecon_hip.so with extra hipcc flags (ablation / A-B builds), next to the product
library:  python scripts/build_alt.py NAME -DSR_WINO_REGV=0 ...  ->  simplerecon_amd/alt/libsr_NAME.so
Select it at run time with SR_HIP_LIBRARY=simplerecon_amd/alt/libsr_NAME.so (simplerecon_amd/_lib.py)."""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplerecon_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    only = None   # --only FILE.hip: recompile just that translation unit with the flags, link the product objects of the rest
    if "--only" in flags:
        i = flags.index("--only")
        only = flags[i + 1]
        flags = flags[:i] + flags[i + 2:]
    bdir = os.path.join("/tmp", "sr_alt_" + name)
    os.makedirs(bdir, exist_ok=True)
    out_dir = os.path.join(ROOT, "simplerecon_amd", "alt")
    os.makedirs(out_dir, exist_ok=True)

    def comp(src):
        if only and os.path.basename(src) != only:
            obj = os.path.join(B.BUILD, os.path.basename(src) + ".o")   # (python -m simplerecon_amd.build first)
            if not os.path.exists(obj):
                raise RuntimeError(f"{obj} missing: build the product library first")
            return obj
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        r = subprocess.run([B.HIPCC] + B.FLAGS + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(comp, B.sources()))
    lib = os.path.join(out_dir, f"libsr_{name}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs +
                          ["-L/opt/rocm/lib", "-lhipblaslt"])
    print(lib)


if __name__ == "__main__":
    main()
