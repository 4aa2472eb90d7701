#!/bin/bash
# usage: bash profiles/r05/build_obj.sh <mode> <name> [extra hipcc flags ..]   ->  build/diag/<name>.o = pt_render_m<mode>.o built with the extra flags,
# through the same four steps as the Makefile's objects (device assembly -> check / repair -> assemble -> embed)
M=$1; NAME=$2; shift; shift
mkdir -p build/diag
make -s OBJDIR=build/diag/$NAME.d EXTRA_HIPFLAGS="$*" build/diag/$NAME.d/pt_render_m$M.o > build/diag/$NAME.log 2>&1 || { tail -5 build/diag/$NAME.log; exit 1; }
mv build/diag/$NAME.d/pt_render_m$M.o build/diag/$NAME.o && rm -rf build/diag/$NAME.d
grep -h "REPAIRED\|DEFECT" build/diag/$NAME.log | cut -c1-200; echo "built build/diag/$NAME.o"
