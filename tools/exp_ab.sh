#!/bin/bash
# generic same-box A/B of one environment switch:  bash tools/exp_ab.sh NAME v1 v2 ...   (headline step;  ELEM=1: also
# the per-element-field variant)
cd $GRAFT_REPO_ROOT
name=$1; shift
for v in "$@"; do
  export $name=$v
  if [ -n "$ELEM" ]; then bash tools/exp_quick.sh ab_${name}_$v elem | sed -n '1p;$p'; else bash tools/exp_quick.sh ab_${name}_$v | head -1; fi
done
