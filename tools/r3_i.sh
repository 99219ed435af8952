mkdir -p gpurun_out
cat > /tmp/probe_dbg.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tools")
import torch
n = int(sys.argv[1])
keep = []
for _ in range(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device="cuda") + 1)
torch.cuda.synchronize()
dev = torch.device("cuda:0")
cand = [torch.cuda.Stream(device=dev) for _ in range(8)]
spin = 600_000
def t(streams):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    for st in streams:
        st.wait_event(e0)
        with torch.cuda.stream(st):
            torch.cuda._sleep(spin)
        cur.wait_stream(st)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
print("single", [round(t([c]), 3) for c in cand])
print("pairs with cand0", [round(t([cand[0], c]), 3) for c in cand[1:]])
print("pairs with cur ", [round(t([torch.cuda.current_stream(), c]), 3) for c in cand])
print("triple 0,1,2", round(t(cand[:3]), 3), "all 8", round(t(cand), 3))
PY
for cfg in "0 d" "3 d" "6 d" "6 8"; do set -- $cfg; if [ "$2" = "8" ]; then export GPU_MAX_HW_QUEUES=8; else unset GPU_MAX_HW_QUEUES; fi; echo "--- dummy=$1 queues=$2"; timeout 120 python /tmp/probe_dbg.py $1 2>&1 | tail -4; done > gpurun_out/r3i_probe_dbg.txt
cat gpurun_out/r3i_probe_dbg.txt
