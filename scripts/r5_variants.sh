# A/B of library variants (roitr_amd/lib/variants/*.so): bash scripts/r5_variants.sh "<command>" name1 name2 ...
export TMPDIR=/tmp
cmd=$1; shift
cp roitr_amd/lib/libroitr_hip.so /tmp/lib_backup.so
for v in "$@"; do
  cp roitr_amd/lib/variants/$v.so roitr_amd/lib/libroitr_hip.so
  echo "== $v"; eval "$cmd" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_backup.so roitr_amd/lib/libroitr_hip.so
