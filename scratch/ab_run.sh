# A/B of product-library variants on the GPU box: bash scratch/ab_run.sh <out-name> "<bench args>" tag1 tag2 ...   (libs from scratch/ab_build.py)
# Every variant is copied over libaq3d.so in turn (the sources are unchanged, so the package does not rebuild) and the bench runs twice per variant, interleaved.
set -e
OUT=gpurun_out/$1; ARGS=$2; shift 2
PKG="quantization-on-3d-object-detection_amd"
cp $PKG/libaq3d.so /tmp/libaq3d_product.so
: > $OUT
for rep in 1 2 3; do
  for tag in "$@"; do
    cp scratch/diag/libaq3d_ab_$tag.so $PKG/libaq3d.so
    python3 bench.py --no-cpu-baseline --no-accuracy $ARGS > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; cp /tmp/libaq3d_product.so $PKG/libaq3d.so; exit 1; }
    python3 -c "
import json,sys
d=json.load(open('/tmp/ab.json'))
print(json.dumps(dict(variant='$tag', rep=$rep, value=d['value'], latency_p50=d['latency_ms_single_frame']['p50'], w16=d.get('value_sparse_w16'), breakdown={k: round(v,3) for k,v in d.get('warmup_breakdown_ms_per_frame',{}).items()})))" >> $OUT
    tail -1 $OUT
  done
done
cp /tmp/libaq3d_product.so $PKG/libaq3d.so
