# per-launch kernel times of every library variant under gpurun_variants/ (tools/build_variant.sh) on ONE box
mkdir -p gpurun_out/r2
for so in gpurun_variants/libvisomatch_*.so; do
  VSM_LIB_PATH=$PWD/$so timeout -k 5 120 python tools/variant_bench.py 2>/dev/null | tr ' ' '\n' | grep -v "^$" | grep "libviso\|exact\|MISMATCH\|^match\|^refine\|^front\|^filters\|^nms\|^emit" | paste -sd' ' || exit 1
done
