#!/bin/bash
# builds experimental variants of libtoyraygun_hip.so into exp_build/<name>/ (timing-only ablations)
set -e
cd /root/repo
name=$1; shift
out=exp_build/$name; mkdir -p $out
C=${SRC:-toyraygun_amd/csrc}
F="-O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fvisibility=hidden -Iinclude -I/root/repo/toyraygun_amd/csrc --offload-arch=gfx950"
# (two translation units per build, as toyraygun_amd/build.py makes them: the path-regeneration kernels apart, without the post-RA scheduler)
R="-mllvm -enable-post-misched=0"
hipcc $F -DTRG_STRICT=0 -DTRG_UNIT=1 "$@" -c $C/trg_kernels.hip -o $out/kf.o &
hipcc $F $R -DTRG_STRICT=0 -DTRG_UNIT=2 "$@" -c $C/trg_kernels.hip -o $out/kf2.o &
hipcc $F -DTRG_STRICT=1 -DTRG_UNIT=1 -ffp-contract=off "$@" -c $C/trg_kernels.hip -o $out/ks.o &
hipcc $F $R -DTRG_STRICT=1 -DTRG_UNIT=2 -ffp-contract=off "$@" -c $C/trg_kernels.hip -o $out/ks2.o &
hipcc $F -x hip "$@" -c $C/trg_capi.cpp -o $out/capi.o &
hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -I/root/repo/toyraygun_amd/csrc --offload-arch=gfx950 "$@" -c $C/trg_build.hip -o $out/build.o &
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libtoyraygun_hip.so $out/kf.o $out/kf2.o $out/ks.o $out/ks2.o $out/capi.o toyraygun_amd/build/bvh_build.o $out/build.o toyraygun_amd/build/trg_group.o -ldl -lpthread
echo built $out
