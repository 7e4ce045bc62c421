cd $GRAFT_REPO_ROOT
o=gpurun_out/r03p; mkdir -p $o
cp garage_amd/libgarage_ec.so /tmp/keep.so
for v in v0 v2_old v1_nobounds v0 v2_old; do
  cp variants/libgarage_ec_$v.so garage_amd/libgarage_ec.so
  echo "== $v" >> $o/ab.txt
  timeout 200 python - >> $o/ab.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, ".")
from tools.host_path_bench import pcie_inclusive_rates
r = pcie_inclusive_rates(512, 5)
print({k.replace("_GiBps", ""): v for k, v in r.items() if k.endswith("GiBps")})
PY
done
cp /tmp/keep.so garage_amd/libgarage_ec.so
cat $o/ab.txt
