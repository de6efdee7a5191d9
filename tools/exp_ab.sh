#!/bin/bash
# generic same-box A/B of one environment switch on the headline step:  bash tools/exp_ab.sh NAME v1 v2 ...
cd $GRAFT_REPO_ROOT
name=$1; shift
for v in "$@"; do
  export $name=$v
  bash tools/exp_quick.sh ab_${name}_$v | head -1
done
