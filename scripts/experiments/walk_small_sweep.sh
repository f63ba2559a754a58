for m in auto 1 0; do
  if [ $m = auto ]; then unset PLVS_WALK_SMALL; else export PLVS_WALK_SMALL=$m; fi
  echo "== walk_small $m"; python scripts/experiments/depth_entry_stream.py 25 100 depth 2>&1 | grep "ms/step"
done
