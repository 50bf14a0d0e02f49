cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multipass or wide16 or work_queue or golden or overflow or high_similarity or full_size or tokens" 2>&1 | tail -5 && \
VARIANTS="base new" EXTRA="--config 5" STEPS=20 bash tools/sweeps/ab.sh
