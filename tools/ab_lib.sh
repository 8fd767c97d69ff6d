#!/bin/bash
# A/B of whole library builds inside one call: usage tools/ab_lib.sh a.so b.so ...   (paths relative to the package dir)
P=open-solution-salt-identification_amd
cp $P/libsaltnet_hip.so /tmp/lib_orig.so
for so in "$@" "$@"; do
  cp $P/$so $P/libsaltnet_hip.so
  echo "== $so"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-iou 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'])"
done
cp /tmp/lib_orig.so $P/libsaltnet_hip.so
