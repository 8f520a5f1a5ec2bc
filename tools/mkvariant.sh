#!/bin/bash
# usage: mkvariant.sh NAME file.hip [extra hipcc flags...]   -> tools/_bin/v/NAME.so (all other objects from the product build)
set -e
NAME=$1; SRC=$2; shift 2
cd /root/repo/paths_amd/csrc
base=$(basename $SRC .hip)
extra=""
if [ "$base" = "attn_x6" ] || [ "$base" = "tlayer_h3" ] || [ "$base" = "tlayer_ws" ]; then extra="-mllvm -amdgpu-mfma-vgpr-form"; fi
hipcc -O3 --offload-arch=gfx950 -fPIC -Wno-unused-value $extra "$@" -c -o /tmp/v_${NAME}.o $(basename $SRC)
objs=""
# (gemm_x6.hip is built in three parts by __graft_entry__.build(): gemm_x6_p1..3.o - a variant replaces all of them by one whole object)
for o in build/*.o; do b=$(basename $o .o); if [ "$b" != "$base" ] && [ "${b%_p[123]}" != "$base" ]; then objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -fPIC -shared -o /root/repo/tools/_bin/v/${NAME}.so $objs /tmp/v_${NAME}.o
echo built $NAME
