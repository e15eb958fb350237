#!/bin/bash
# Upper bound on what a wide-network training step could gain from a dX chain that recomputes Y_l = W_l X_l from the X dumps
# instead of reading Y dumps (16 KB per sample instead of 19): timing builds whose forward writes no Y dumps (-DRESNET_AB_NO_Y) and
# whose chain issues every W^T dZ tile's MFMAs twice (-DRESNET_AB_2X: the recomputation's matrix work without its second weight
# stream and operand registers).  Through gpurun, from the repo root, after
#   tools/ab/build_variant.sh wide_noy "-DRESNET_AB_NO_Y" so3x_resnet.hip
#   tools/ab/build_variant.sh wide_2x "-DRESNET_AB_2X" so3x_resnet.hip
#   tools/ab/build_variant.sh wide_noy_2x "-DRESNET_AB_NO_Y -DRESNET_AB_2X" so3x_resnet.hip
# The builds compute garbage gradients; only their times count.  The library is swapped on the box's scratch copy only.
R=$GRAFT_REPO_ROOT
lib=$R/diffusion-extensions_amd/libso3x.so
cp $lib /tmp/libso3x_product.so
for round in 1 2; do
  for v in product wide_noy wide_2x wide_noy_2x; do
    if [ $v = product ]; then cp /tmp/libso3x_product.so $lib; else cp $R/build/libso3x_$v.so $lib; fi
    python3 $R/tools/ab/ab_wide_train.py $v 2>&1 | grep fwd_stash
    python3 $R/tools/ab/ab_wide_step.py $v 2>&1 | tail -1
  done
done
cp /tmp/libso3x_product.so $lib
