# multi-pass fill in buffer form: parity first (multi-pass, wide, work-queue tests), then A/B on configs 4 and 5
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multipass or wide16 or work_queue or golden or overflow or high_similarity or full_size" 2>&1 | tail -8 && \
VARIANTS="base new" EXTRA="--config 4" STEPS=6 bash tools/sweeps/ab.sh && \
VARIANTS="base new" EXTRA="--config 5" STEPS=20 bash tools/sweeps/ab.sh
