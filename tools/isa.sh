#!/bin/bash
# Device ISA of the library (gfx950) -> /tmp/isa/dsgd.s; prints register/occupancy lines of kernels matching $1
set -e
mkdir -p /tmp/isa; rm -f /tmp/isa/_Z*.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -std=c++17 -w -S --cuda-device-only \
  -I"$(dirname "$0")/../include" -o /tmp/isa/dsgd.s "$(dirname "$0")/../distributed-sgd_amd/csrc/dsgd_hip.hip" 2>&1 | grep -v "warning" | grep -B2 -A6 " error" || true
pat="${1:-wseg}"
for k in $(grep -o "^_Z[A-Za-z0-9_]*${pat}[A-Za-z0-9_]*:" /tmp/isa/dsgd.s | tr -d ':' | sort -u); do
  awk -v k="$k" '$0 ~ "^"k":" {p=1} p {print} p && /\.end_amdhsa_kernel/ {exit}' /tmp/isa/dsgd.s > /tmp/isa/$k.s
  # the metadata comment block follows .end_amdhsa_kernel
  awk -v k="$k" '$0 ~ "^"k":" {p=1} p && /; (NumVgprs|NumSgprs|ScratchSize|Occupancy):/ {print} p && /; Occupancy:/ {exit}' /tmp/isa/dsgd.s | tr '\n' ' '
  echo " <- $(echo $k | c++filt | cut -c1-60)"
done
