#!/bin/bash
# builds tools/vmm_probe (here, cross-compiled) or runs it N times (GPU box):  tools/vmm_probe.sh build | run [N] [align]
cd "$(dirname "$0")/.." || exit 1
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/vmm_probe.hip -o tools/vmm_probe -Lgst-plugins-bad_amd -lmibayer -Wl,-rpath,'$ORIGIN/../gst-plugins-bad_amd'
else
  for i in $(seq 1 "${2:-3}"); do tools/vmm_probe ${3:-} 2>&1; done
fi
