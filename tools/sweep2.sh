for dbg in 0 1 2; do echo "DEBUG=$dbg"; SPATTEN_DEBUG=$dbg PSPLITS=8,16 python tools/probe_sweep.py 2>&1 | tail -3; done
