import sys, os, glob, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in sorted(glob.glob(os.path.join(root, 'build_ablate', 'libmpcx_*.bin'))):
    shutil.copy(f, os.path.join(root, 'mpc_for_av_at_intersection_amd', 'libmpcx.so'))
    out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'qp_timing.py'), '20', '32768'], capture_output=True, text=True).stdout.strip().splitlines()
    print(os.path.basename(f), out[-1] if out else 'no output')
