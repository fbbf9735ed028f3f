ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_prior
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_prior -o p -- python $ROOT/scripts/prior_overhead.py $1 $2 > /tmp/ks_prior.log 2>&1
tail -2 /tmp/ks_prior.log
f=$(find /tmp/ks_prior -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print("%-58s calls %5s avg %8.1f us  %5s%%" % (r["Name"].split("(")[0][:58], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
