for t in 0 1e-10 1e-8 1e-6 1e-4 1e-3; do
  echo "== VINTERP_WALK_TOL=$t"
  VINTERP_WALK_TOL=$t python tools/exp_walk_floor.py 1000 2>&1 | tail -5
  VINTERP_WALK_TOL=$t python tools/perf_fit.py 1000 2>&1 | cut -c1-90
done
