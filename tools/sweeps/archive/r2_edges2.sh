# scheduler strategies for the kernels file (whole library variants), multi-pass and single-pass shapes
cd $GRAFT_REPO_ROOT
VARIANTS="base new ilp trk minreg" EXTRA="--config 4" STEPS=6 bash tools/sweeps/ab.sh && \
VARIANTS="base new ilp trk minreg" EXTRA="--config 5" STEPS=20 bash tools/sweeps/ab.sh && \
VARIANTS="base new ilp trk minreg" EXTRA="--config 2" STEPS=100 bash tools/sweeps/ab.sh
