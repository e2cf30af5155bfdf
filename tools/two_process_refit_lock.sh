#!/bin/bash
# two chol_soak instances at once on one device: time-outs per setting of the cross-process refit lock
for setting in "BOHIP_DF_FILE_LOCK=300" "BOHIP_DF_FILE_LOCK=60000" "BOHIP_DF_FILE_LOCK=0"; do
  echo "== $setting"
  (env $setting timeout 300 python tools/chol_soak.py 3000 6 8 2>&1 | grep -E "saw a time-out|timed out" | sed 's/^/A: /' &)
  env $setting timeout 300 python tools/chol_soak.py 3000 6 8 2>&1 | grep -E "saw a time-out|timed out" | sed 's/^/B: /'
  wait; sleep 2
done
ls -la /tmp/bohip-* /run/user 2>&1 | head; echo XDG=$XDG_RUNTIME_DIR
