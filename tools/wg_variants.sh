# usage: bash tools/wg_variants.sh "variants ('base' = the shipped library)" "workloads" "wave settings"
for v in $1; do
  echo "### variant $v"
  lib=$PWD/compress_amd/lib/libxflate_amd_$v.so
  [ "$v" = base ] && lib=$PWD/compress_amd/lib/libxflate_amd.so
  XFG_LIB=$lib bash tools/wg_bench.sh "$2" "$3"
done
