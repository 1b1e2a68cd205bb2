for T in 1000 4000; do for p in 1 2 3 4 6; do echo "T=$T pipelines=$p: $(VINTERP_PIPELINES=$p python tools/perf_fit.py $T 2>&1 | cut -c38-90)"; done; done
