#!/bin/bash
cd $GRAFT_REPO_ROOT
for tr in 0 4:4 5:5 8:8 6:4 6:8; do
  if [ "$tr" == "0" ]; then unset DIFFHE_FUSED_TR_PRE; else export DIFFHE_FUSED_TR_PRE=$tr; fi
  bash tools/exp_quick.sh r4m_tr${tr/:/_}
done
