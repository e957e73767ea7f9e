cd $GRAFT_REPO_ROOT
for n in 100000 500000 1500000 3000000; do
  timeout 600 python bench.py --anchors $n --steps 12 --warmup 9 --no-cpu-baseline --no-heavy --no-eval-fps --no-image-loss 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['codec']; cf=d['config']
v2=c.get('container_v2',{})
print($n, cf['gaussians_per_view'], cf['tile_pairs_per_view'], d['value'], d['ms_per_step'], d['value_raster_only'], '| v1', c['encode_Manchors_per_s'], c['decode_Manchors_per_s'], c['bitstream_MB'], c['decoded_anchor_and_masks_bit_exact'], '| v2', v2.get('encode_Manchors_per_s'), v2.get('decode_Manchors_per_s'), v2.get('bitstream_MB'), v2.get('decoded_anchor_and_masks_bit_exact'))"
done
