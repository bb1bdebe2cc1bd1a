cd $GRAFT_REPO_ROOT
for o in 1 0 1 0; do echo "== ASM_BN_ORDER=$o"; ASM_BN_ORDER=$o timeout 120 python tools/bn_bench.py 2>&1 | tail -14; done
