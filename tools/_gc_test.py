import gc, json, os, subprocess, sys
mode = sys.argv[1]
sys.argv = ['bench.py', '--legs', 'spectrum', '--no-cpu', '--no-config5']
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ['WBX_BENCH_COMPOSITE_RUNS'] = '16'
if mode == 'nogc':
  gc.disable()
elif mode == 'freeze':
  gc.freeze()
import bench
bench.main()
