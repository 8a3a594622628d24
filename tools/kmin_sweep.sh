#!/bin/bash
# opening time (16 rounds etc.) against the minimum task length of the accumulation, per SRS size: the data behind the task-length rule of msm.hip
for sz in 14 15 16 17 18; do
  line="2^$sz:"
  for k in 4 5 6 7 8 9 10 11 12 13 14 16 20; do
    t=$(KH_KMIN=$k python tools/ipa_time.py $sz 2>&1 | grep '^rep 2' | sed 's/.*rounds \([0-9.]*\) ms.*/\1/')
    line="$line $k=$t"
  done
  echo "$line"
done
