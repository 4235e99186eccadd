#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters...>" -- <command...>     (run on the GPU box; output under gpurun_out/pmc_<tag>/)
tag=$1; ctrs=$2; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd $root
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- "$@" > $out.log 2>&1
echo "pmc $tag exit $?"
python3 $root/tools/pmc_summary.py $out 2>&1 | cut -c1-400 | head -${PMC_LINES:-12}
