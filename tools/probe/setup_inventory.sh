for D in 0 0x1000000 0x2000000 0x4000000 0x8000000 0x10000000; do
  echo "--- PG_DEBUG=$D"
  PG_LIB=procgen2_amd/lib/libprocgen2_hip_ablate.so PG_DEBUG=$D PG_SETTLE=100 PG_KERNEL=setup_kernel tools/pmc_quick.sh r04e_$D "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" 2>&1 | grep "per wave"
done
