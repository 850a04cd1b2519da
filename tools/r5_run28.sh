for i in 1 2; do
  timeout 1200 python -m pytest tests -q -m gpu -rfs -k "not soak and not fullsize" 2>&1 | grep -E "passed|failed|FAILED|rc=|SKIPPED" | cut -c1-200 | head -12
done
