import sys; sys.path.insert(0,'/root/repo'); sys.argv=['x']
exec(open('/root/repo/tools/gemm_probe.py').read().split("if __name__")[0])
for N in (80,160,320):
    for tile in (0,1):
        us,tf=time_nt(153600,N,1536,tile)
        print("NT M=153600 N=%d K=1536 tile=%d: %.1f us %.0f TF/s" % (N,tile,us,tf))
for (N,K) in ((512,512),(1536,80),(1536,160)):
    us,tf=time_tn(153600,N,K)
    print("TN M=153600 N=%d K=%d: %.1f us %.0f TF/s" % (N,K,us,tf))
