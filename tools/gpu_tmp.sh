export SONET_HIP_LIB=so-net_amd/lib/libsonet_hip_variants.so
for r in 8 6 12 4; do echo "== SONET_IM_R=$r"; SONET_IM_R=$r python tools/microbench.py index_max 2>&1 | grep "f32 B=64 C=384 N'=15000\|f32 B=8 \|bf16 B=64 C=384 N'=15000"; done
