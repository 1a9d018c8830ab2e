#!/bin/bash
# usage: tools/run_variants.sh <tool.py and args> -- name1 name2 ...   (runs the tool once per variant library)
set -o pipefail
cmd=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do cmd+=("$1"); shift; done
shift
for v in "$@"; do
  echo "== $v"
  MVDB_HIP_LIB=$PWD/minivectordb_amd/lib/libmvdb_hip_$v.so timeout -k 10 200 python "${cmd[@]}" || exit $?
done
