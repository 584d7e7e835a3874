#!/bin/bash
# like tools/try_cfg_variants.sh, but with the oracle comparison of the listed configurations (a variant must stay exact)
# usage: tools/try_cfg_parity.sh <configs> <oracle-configs> [names...]
set -u
cd "$(dirname "$0")/.."
CFG=$1; ORC=$2; shift 2
cp rusterix_amd/csrc/librxr_hip.so /tmp/librxr_hip_orig.so
trap 'cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so' EXIT
for name in "$@"; do
  if [ "$name" = base ]; then cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so; else cp "build/variants/librxr_hip_$name.so" rusterix_amd/csrc/librxr_hip.so; fi
  echo "== $name"
  timeout 600 python tools/run_configs.py --configs "$CFG" --oracle "$ORC" --frames 20 > /tmp/cfg_$name.jsonl 2>&1
  python3 tools/show_cfg.py /tmp/cfg_$name.jsonl
done
