#!/bin/bash
# Build a variant of libclipa_hip.so with extra -D flags on ONE source (A/B experiments with tools/gemm_lib_ab.py):
#   tools/build_variant.sh NAME SOURCE.hip -DFOO=1 ...   ->  clipa_amd/lib/libclipa_var_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m clipa_amd.build > /dev/null
obj=clipa_amd/lib/obj/var_${name}.o
# SOURCE: a file of clipa_amd/csrc/ (replaces the library object of the same name) or a path such as tools/probes/x.hip (added)
if [ -f "clipa_amd/csrc/$src" ]; then path=clipa_amd/csrc/$src; else path=$src; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I clipa_amd/csrc -I include -Wno-unused-result -ffp-contract=fast "$@" -c $path -o $obj
# the objects of build.SOURCES only (the object directory also holds -save-temps by-products of the audited kernels)
others=$(python -c "import clipa_amd.build as b; print(' '.join('clipa_amd/lib/obj/' + s[:-4] + '.o' for s in b.SOURCES if s != '$src'))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o clipa_amd/lib/libclipa_var_${name}.so $obj $others
echo clipa_amd/lib/libclipa_var_${name}.so
