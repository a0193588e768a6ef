cd $GRAFT_REPO_ROOT
run() { out=$(env DMP_VGRU_DETACH=$1 python bench.py --no-cpu-baseline --no-exact-f32 --steps 5 --warmup 1 2>/dev/null)
  python3 - "DMP_VGRU_DETACH=$1" "$out" <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2].strip().splitlines()[-1])
    print("%-28s %.3f structures/s  chip_ms/launch %.4f  in flight %.2f  ok %s" % (sys.argv[1], j["value"], j["roofline"]["chip_ms_per_launch"], j["roofline"]["launches_in_flight"], j["verify"]["ok"]))
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][-300:])
PY
}
for i in 1 2; do run 0; run 2; done
