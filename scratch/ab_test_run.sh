# bash scratch/ab_test_run.sh <variant-tag> "<pytest args>" <out-name> "<bench args>" tag...: pytest on one variant of scratch/ab_build.py, then the A/B of scratch/ab_run.sh
PKG="quantization-on-3d-object-detection_amd"
cp $PKG/libaq3d.so /tmp/libaq3d_product0.so
cp scratch/diag/libaq3d_ab_$1.so $PKG/libaq3d.so
python3 -m pytest tests/test_sparse_gpu.py tests/test_fullsize_gpu.py -x -q -k "$2" > gpurun_out/ab_test_$1.log 2>&1; rc=$?
tail -3 gpurun_out/ab_test_$1.log
cp /tmp/libaq3d_product0.so $PKG/libaq3d.so
[ $rc -ne 0 ] && exit $rc
shift 2
bash scratch/ab_run.sh "$@"
