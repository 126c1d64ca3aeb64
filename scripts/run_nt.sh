cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_mlp_volume.py -q -x 2>&1 | tail -1
for x in 0 1; do
  tag=xcd$x
  for wl in hero_cfg5_volume hero_cfg3_volume; do
  SR_MLP_XCD=$x python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag $wl', round(d['ms_per_step'],3))"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp; rm -rf $O/pmcnt_${c}_$tag; SR_MLP_XCD=$x timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmcnt_${c}_$tag -o x -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1)
  done
  python - <<PY
import csv, glob, collections
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for f in glob.glob("$O/pmcnt_%s_$tag/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "sr_mlp_volume_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); n += 1
    v[c] = tot / max(n, 1)
print("$tag $wl FETCH KiB", round(v["FETCH_SIZE"]), "WRITE KiB", round(v["WRITE_SIZE"]), "HBM-side MB (2F+W)", round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e6, 1))
PY
  done
done
