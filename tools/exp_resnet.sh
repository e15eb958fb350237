#!/bin/bash
# timing experiments for the resnet chain (development aid)
for e in 0 1 2 3 4 5 6 7; do
  echo "EXP=$e"; SO3X_RESNET_EXP=$e python tools/kbench.py resnet 2>&1 | grep resnet_chain | tail -1
done
