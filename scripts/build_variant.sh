#!/bin/bash
# Build an alternate of libndsm_hip.so in which ONE translation unit is compiled with extra -D flags (dev aid):
#   scripts/build_variant.sh <name> <unit> [-DFOO=1 ...]     -> scripts/bin/libv_<name>.so
#   scripts/build_variant.sh <name> <unit>@<git-rev>          -> that unit's source taken from a revision
# Run the timing scripts against it with NDSM_HIP_LIB=scripts/bin/libv_<name>.so.
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
rev=""
case "$unit" in *@*) rev=${unit#*@}; unit=${unit%@*};; esac
make -C ndsm_amd -j8 >/dev/null
mkdir -p scripts/bin/obj_$name
src=ndsm_amd/csrc/$unit.hip
if [ -n "$rev" ]; then git show "$rev:ndsm_amd/csrc/$unit.hip" > scripts/bin/obj_$name/$unit.hip; src=scripts/bin/obj_$name/$unit.hip; fi
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Indsm_amd/csrc "$@" -c $src -o scripts/bin/obj_$name/$unit.o
objs=""
for o in ndsm_amd/build/*.o; do
  if [ "$(basename $o)" = "$unit.o" ]; then objs="$objs scripts/bin/obj_$name/$unit.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/amdflang -shared $objs -o scripts/bin/libv_$name.so -Wl,--version-script=ndsm_amd/exports.map \
  -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -lrccl -lstdc++ -lm
echo built scripts/bin/libv_$name.so
