#!/bin/bash
# Same-box A/B of bench.py variants, interleaved repetitions (run via gpurun).  ONE parametrised script instead of one per experiment:
#   tools/ab.sh OUTDIR [-r REPS] "label|bench args" "label|bench args" ...
# bench args select what differs: --dtype / --masks / --mode, --lib <variant .so> (tools/build_variant.sh NAME "-D..."), --set nets.FUSE_BNBWD=True
# (plan-compiler switches).  Example (round 2's CTL_FUSE_BNBWD A/B):
#   tools/ab.sh gpurun_out/ab_bnbwd "default|" "fused|--set nets.FUSE_BNBWD=True"
# Prints one line per run and a median table; raw JSON lines land in OUTDIR.
out=$1; shift
reps=2
if [ "$1" = "-r" ]; then reps=$2; shift; shift; fi
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p $out
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    label=${v%%|*}; args=${v#*|}
    timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub-records $args > $out/${label}_$rep.json 2> $out/${label}_$rep.err
    python3 - "$out/${label}_$rep.json" "$label" "$rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-24s rep %s: %8.1f slices/s %7.3f ms/step  mode %s  launches %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d.get("mode"), d.get("launches_per_step", {}).get("library")))
except Exception as e:
    print(sys.argv[2], "rep", sys.argv[3], "FAILED", e)
    print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
  done
done | tee $out/ab.txt
python3 - "$out" "$@" <<'PY' | tee -a $out/ab.txt
import glob, json, statistics, sys
out = sys.argv[1]
print("---- medians")
for v in sys.argv[2:]:
    label = v.split("|")[0]
    ms = []
    for f in sorted(glob.glob(f"{out}/{label}_*.json")):
        try: ms.append(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"])
        except Exception: pass
    if ms: print("%-24s median %.3f ms/step over %d runs (min %.3f max %.3f)" % (label, statistics.median(ms), len(ms), min(ms), max(ms)))
PY
