R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
cp minimap2_amd/libmm2amd.so /tmp/main.so
for v in combopad combo; do cp minimap2_amd/build/variants/libmm2amd_$v.so minimap2_amd/libmm2amd.so; echo "variant $v"; timeout 60 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "test_ont_sam_identical" 2>&1 | tail -1; done
cp /tmp/main.so minimap2_amd/libmm2amd.so
