#!/bin/bash
# Driver-literal GPU check: the two commands the round-end driver runs (pytest -m gpu -x, then smoke()), with NO torch.cuda call in
# front of the library's first copy, plus the box facts that tell a lease-level fault from a product fault.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_fresh.sh TAG [pytest args]'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
TAG="${1:-fresh}"; shift
O=gpurun_out/${TAG}
mkdir -p gpurun_out
{
  echo "== box"; date -u
  cat /opt/rocm/.info/version 2>/dev/null
  cat /sys/module/amdgpu/version 2>/dev/null | sed 's/^/amdgpu module: /'
  uname -r
  echo "ulimit -l: $(ulimit -l)   HSA_XNACK=${HSA_XNACK:-unset}  HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-unset}"
  nproc; free -g | head -2
  /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|Node:|Name: +gfx|Uuid|Compute Unit|KERNEL_DISPATCH|Max Clock" | head -40
  ls /dev/dri /dev/kfd 2>&1 | tr '\n' ' '; echo
} > ${O}_box.log 2>&1
echo "== pytest (driver literal)" 
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider "$@" > ${O}_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 ${O}_pytest.log
echo "== smoke (driver literal)"
timeout 600 python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e
e.smoke(); print("__SMOKE_OK__")' > ${O}_smoke.log 2>&1; echo "smoke rc=$?"
tail -4 ${O}_smoke.log
