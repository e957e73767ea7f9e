# A/B on ONE box: alternate two environment settings, N rounds each; usage: tools/ab.sh "ENVA=1" "ENVB=1" [rounds]
A="$1"; B="$2"; R=${3:-3}
run() { env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-codec --no-raster-only --no-image-loss 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in $(seq $R); do run "$A"; run "$B"; done
