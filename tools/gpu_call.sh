R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout 45 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_ksw.py -x -q -m gpu -k "ont_sam or hifi_sam or streaming" 2>&1 | tail -2
