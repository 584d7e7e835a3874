#!/bin/bash
# C5 end to end (tools/e2e_probe.py) for several numbers of streaming groups (RXR_STREAM_GROUPS).  usage: tools/stream_groups_probe.sh 8 16 32
cd "$(dirname "$0")/.."
for g in "$@"; do
  RXR_STREAM_GROUPS=$g python tools/e2e_probe.py --config C5 --frames 20 2>/dev/null | head -1 > /tmp/sg.json
  python - "$g" <<'PY'
import json, sys
d = json.load(open("/tmp/sg.json"))
print("groups", sys.argv[1], "project", d["project_ms"], "project+handover", d["project_plus_handover_ms"], "render+download", d["render_download_ms"], "whole call", d["whole_call_ms"])
PY
done
