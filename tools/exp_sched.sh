for s in 3e-2 1e-2 3e-3 1e-3 1e-1 "1e-1,1e-3" "3e-2,1e-4"; do
  echo "== schedule $s"
  VINTERP_REBASE_SCHEDULE=$s python tools/exp_brent_stamps.py 1000 2>&1 | grep -E "fit |sweeps per|busy"
  VINTERP_REBASE_SCHEDULE=$s python tools/perf_fit.py 1000 2>&1 | cut -c1-80
done
