cd /tmp; export TMPDIR=/tmp CVR_NO_FUSED=1
for args in "nodict" "f32"; do for lds in 0 1; do
  if [ $lds = 1 ]; then export CVR_CONVERT_LDS=1; else unset CVR_CONVERT_LDS; fi
  rm -rf /tmp/ks; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $GRAFT_REPO_ROOT/tools/wg_create_once.py $args > /dev/null 2>&1
  python3 -c "
import csv,glob
for f in glob.glob('/tmp/ks/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'convert' in r['Name']: print('$args lds=$lds', r['Name'][:60], r['Calls'], r['AverageNs'])"
done; done
