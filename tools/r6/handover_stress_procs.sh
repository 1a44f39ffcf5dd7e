#!/bin/bash
# PROCS copies of ONE variant of tools/exp/handover_stress at once (the configuration of the one sighting had eight
# processes on the GPU): tools/r6/handover_stress_procs.sh OUTDIR SECS PROCS -- <stress options>
out=$1; secs=$2; procs=$3; shift 4
mkdir -p "$out"
bin=tools/exp/handover_stress
[ -x $bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -pthread tools/exp/handover_stress.hip -o $bin || exit 2
for p in $(seq 1 "$procs"); do
  timeout $((secs + 120)) $bin --seconds "$secs" "$@" > "$out/proc$p.json" 2> "$out/proc$p.err" &
done
wait
cat "$out"/proc*.json
python3 - "$out" <<'PY'
import glob, json, sys
tot = bad = 0
for p in glob.glob(sys.argv[1] + "/proc*.json"):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        continue
    tot += d["handovers"]
    bad += d["h2d_stream_check"]["wrong_words"] + d["compute_stream_check"]["wrong_words"]
print(json.dumps({"processes": len(glob.glob(sys.argv[1] + '/proc*.json')), "handovers_total": tot, "wrong_words_total": bad}))
PY
