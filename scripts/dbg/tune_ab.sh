# sampler and train throughput with warm-L2 and cold-L2 tuning, alternating, same box
for c in 0 1 0 1; do echo "== MADIFF_TUNE_COLD=$c"; MADIFF_TUNE_COLD=$c timeout -k 10 300 python bench.py --no-dp1 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), round(d['sample']['value'],1))" || exit 1; done
