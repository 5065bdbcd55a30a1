set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05_8; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_p2p_exchange.py tests/test_kernels.py -m gpu -x -q -k "p2p or dwconv" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
timeout 600 python tools/exchange_overlap.py b_nus_bn > $O/r05_grad_exchange_overlap_b_nus.txt 2>$O/xo.err; echo "overlap b rc $?" | tee -a $O/summary.txt; cat $O/r05_grad_exchange_overlap_b_nus.txt | tee -a $O/summary.txt
timeout 600 python tools/exchange_overlap.py t_nus_bn 4 > $O/r05_grad_exchange_overlap_t_nus_forced4.txt 2>>$O/xo.err; echo "overlap t rc $?" | tee -a $O/summary.txt; tail -12 $O/r05_grad_exchange_overlap_t_nus_forced4.txt | tee -a $O/summary.txt
tail -5 $O/xo.err
