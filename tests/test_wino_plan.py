"""The Winograd launch plan (csrc/sr_wino.hip: sr_wino_plan -- channel block NT and split-K factor per layer shape) is a
host-side cost model fitted on measurements: profiles/r03_wino_plan_sweep.txt holds every 3x3 shape of the hero conv stack
at batch 8 and 1 timed under each forced plan on an MI355X.  The model must keep choosing a plan whose MEASURED time is
close to the best measured one (runs without a GPU: the plan functions are plain host code of the C-ABI library)."""
import os
import re

from simplerecon_amd import _lib

TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_wino_plan_sweep.txt")


def _rows():
    header, rows = None, []
    for line in open(TABLE):
        if line.startswith("(B,Ci,H,W,Co)"):
            header = [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"nt(\d)ks(\d)", line.split("best")[0])]
            continue
        m = re.match(r"\((\d+), (\d+), (\d+), (\d+), (\d+)\)\s+(.*)", line)
        if m and header:
            vals = m.group(6).split()
            times = [float(v) for v in vals[1:1 + len(header)]]          # vals[0] = the default plan of that day
            rows.append((tuple(int(g) for g in m.groups()[:5]), dict(zip(header, times))))
    return rows


def test_plan_table_is_present_and_complete():
    rows = _rows()
    assert len(rows) == 46 and all(len(t) == 8 for _, t in rows)


def test_chosen_plan_is_close_to_the_best_measured_one():
    lib = _lib.lib()
    total_chosen = total_best = 0.0
    worst = 0.0
    for (b, ci, h, w, co), times in _rows():
        ks = lib.sr_wino_splitk_factor(b, h, w, ci, co)
        nt = int(re.search(r"<(\d),", lib.sr_wino_kernel_name(b, h, w, ci, co, 1, 1).decode()).group(1))
        assert nt in (1, 2) and ks in (1, 2, 4, 8)
        chosen, best = times[(nt, ks)], min(times.values())
        total_chosen += chosen
        total_best += best
        if best >= 30.0:   # below ~30 us a launch is latency: the table's run-to-run noise there is ~8 %
            worst = max(worst, chosen / best)
    # per shape within 6 % of the best measured plan, 3 % over all shapes and both batch sizes
    assert worst < 1.06, worst
    assert total_chosen < 1.03 * total_best, (total_chosen, total_best)
