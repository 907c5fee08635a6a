cd $GRAFT_REPO_ROOT
VARIANTS="cur g4w4 g4w2 g4w1 g4w4x16 g4w4x0 g4w4o7" WLS="target c2" SEGS="0" ROUNDS=3 bash tools/ab_variants.sh
