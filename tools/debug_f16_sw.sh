#!/bin/bash
for e in X=1 DMME_NO_WS=1 DMME_NO_CONV1X1_AS=1 DMME_NO_DGRAD_DIRECT=1 DMME_NO_RES_ALIAS=1 DMME_NO_RES_EXTRA=1 DMME_NO_GN_BWD_ROWS=1 DMME_NO_GN_BWD_FUSED_FIN=1 DMME_NO_COLSUM_GROUP=1 DMME_NO_BIAS_GROUP=1 DMME_NO_GN_IN=1 DMME_NO_FUSED_GN=1 DMME_NO_LVL=1 DMME_NO_ATTN_FULL=1; do
  echo "== $e"; env $e python tools/debug_f16_grads.py ${1:-128} fp16 2>/dev/null | grep "up_layers.13.conv2.2.bias\|up_layers.10.conv2.2.bias\|down_layers.0.conv1.2.bias\|up_layers.12.conv2.2.bias"
done
