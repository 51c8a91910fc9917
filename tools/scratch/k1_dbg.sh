for m in 0 1 2 3 4 8 15; do echo "VO_DBG_K1=$m"; VO_DBG_K1=$m timeout -k 5 60 python tools/scratch/k1_probe.py 2>&1 | grep -E "nms_candidates"; done
