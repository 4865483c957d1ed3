#!/bin/bash
# each probe in its own process; a crash of one does not stop the rest (they are independent and tiny)
for args in "all 1 big eager drop" "all 1 big eager zero nosync drop"; do
  timeout -k 5 120 python scripts/capture_probe.py $args 2>&1 | grep -E "^OK|Segmentation|Error|error" | head -3
  echo "exit[$args]=${PIPESTATUS[0]}"
done
