#!/bin/bash
# usage: bash profiles/r05/with_objs.sh "<mode>=<object> [<mode>=<object> ..] [api=<pt_api object>]" <command ...>
# Runs <command> with a library in which the render object of each named mode (pt_render_m<mode>.o) is replaced by the given object
# (built here beforehand with profiles/r05/build_obj.sh), linked on the GPU box; the shipped library is put back afterwards.
SPEC=$1; shift
cp portrayer_amd/libportrayer_hip.so /tmp/keep.so
O=portrayer_amd/csrc
API=$O/pt_api.o; for kv in $SPEC; do if [ "${kv%%=*}" = "api" ]; then API="${kv#*=}"; fi; done; objs="$API $O/pt_build.o $O/pt_node.o"
for m in 1 2 3 4 5 6 7 8 9; do
  o="$O/pt_render_m$m.o"
  for kv in $SPEC; do if [ "${kv%%=*}" = "$m" ]; then o="${kv#*=}"; fi; done
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs -o portrayer_amd/libportrayer_hip.so -ldl || { echo "link failed"; cp /tmp/keep.so portrayer_amd/libportrayer_hip.so; exit 1; }
"$@"
rc=$?
cp /tmp/keep.so portrayer_amd/libportrayer_hip.so
exit $rc
