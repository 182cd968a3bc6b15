set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/tapdh_probe.py 2>&1 | grep -v amdgpu | grep "fwd"
SSBEV_TAPDH_GPC=9 timeout 300 python tools/tapdh_probe.py 2>&1 | grep -v amdgpu | grep "fwd  hint 0" | sed "s/^/gpc9 /"
SSBEV_TAPDH_GPC=23 timeout 300 python tools/tapdh_probe.py 2>&1 | grep -v amdgpu | grep "fwd  hint 0" | sed "s/^/gpc23 /"
