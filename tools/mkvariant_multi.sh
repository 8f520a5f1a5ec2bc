#!/bin/bash
# usage: mkvariant_multi.sh NAME "extra hipcc flags" file1.hip [file2.hip ...]  -> tools/_bin/v/NAME.so
# (the named sources recompiled with the extra flags, every other object from the product build; gemm_x6.hip as one whole object)
set -e
NAME=$1; FLAGS=$2; shift 2
cd /root/repo/paths_amd/csrc
skip=""; new=""
for SRC in "$@"; do
  base=$(basename $SRC .hip)
  extra=""
  if [ "$base" = "attn_x6" ] || [ "$base" = "tlayer_h3" ] || [ "$base" = "tlayer_ws" ]; then extra="-mllvm -amdgpu-mfma-vgpr-form"; fi
  hipcc -O3 --offload-arch=gfx950 -fPIC -Wno-unused-value $extra $FLAGS -c -o /tmp/v_${NAME}_${base}.o ${base}.hip &
  skip="$skip $base"; new="$new /tmp/v_${NAME}_${base}.o"
done
wait
objs=""
for o in build/*.o; do b=$(basename $o .o); b=${b%_p[123]}; keep=1; for s in $skip; do [ "$b" = "$s" ] && keep=0; done; [ $keep = 1 ] && objs="$objs $o"; done
mkdir -p /root/repo/tools/_bin/v
hipcc --offload-arch=gfx950 -fPIC -shared -o /root/repo/tools/_bin/v/${NAME}.so $objs $new
echo built $NAME
