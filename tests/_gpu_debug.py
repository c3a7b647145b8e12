import sys, torch, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g
g.smoke()
