"""The measurement helpers under tools/ that can run without a GPU (they are what profiles/ is made with)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_by_grid_aggregates_and_measures_overlap(tmp_path):
    trace = tmp_path / 'p_kernel_trace.csv'
    head = ('"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id",'
            '"Start_Timestamp","End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count",'
            '"Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"\n')
    rows = []
    t = 1000
    for step in range(2):
        # two launches of kernel A one after the other (1 ms each), kernel B overlapping the second one by 0.4 ms
        rows.append(('void kA<64, 128>(ConvP)', t, t + 1000000, 512, 512 * 256, 1, 1)); t += 1000000
        rows.append(('void kA<64, 128>(ConvP)', t, t + 1000000, 512, 512 * 256, 1, 1))
        rows.append(('kB(float*)', t + 600000, t + 1600000, 256, 256 * 8, 1, 1)); t += 2000000
    with open(trace, 'w') as f:
        f.write(head)
        for i, (name, t0, t1, wg, gx, gy, gz) in enumerate(rows):
            f.write('"KERNEL_DISPATCH","Agent 2",1,0,1,%d,1,"%s",%d,%d,%d,0,0,8,0,32,%d,1,1,%d,%d,%d\n'
                    % (i, name, i, t0, t1, wg, gx, gy, gz))
    out = tmp_path / 'agg.jsonl'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'trace_by_grid.py'), str(trace), '--steps', '2', '--out',
                        str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = [json.loads(l) for l in open(out)]
    summary, kernels = recs[0], recs[1:]
    assert abs(summary['overlapped_ms_per_step'] - 0.4) < 1e-6 and abs(summary['gpu_busy_ms_per_step'] - 2.6) < 1e-6
    a = next(k for k in kernels if k['kernel'].startswith('void kA'))
    assert a['launches_per_step'] == 2.0 and a['avg_us'] == 1000.0 and a['workgroups'] == 256
    b = next(k for k in kernels if k['kernel'].startswith('kB'))
    assert b['launches_per_step'] == 1.0 and b['workgroups'] == 8
