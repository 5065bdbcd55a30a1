import torch, time
dev=torch.device('cuda:0')
x=torch.zeros(1024,device=dev)
def work(n, t):
    for _ in range(n): t.add_(1.0)
def host_time(fn, n=20):
    torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        torch.cuda.synchronize(); t0=time.perf_counter(); fn(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        ts.append(((t1-t0)*1e3,(t2-t0)*1e3))
    ts.sort(); return ts[len(ts)//2]
def capture(body):
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    return g
N=1000
a=torch.zeros(1024,device=dev); b=torch.zeros(1024,device=dev)
g1=capture(lambda: work(N,a))
print('linear %d nodes: host %.3f ms total %.3f ms'%((N,)+host_time(g1.replay)))
s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
def two():
    cur=torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): work(N//2,a)
    with torch.cuda.stream(s2): work(N//2,b)
    cur.wait_stream(s1); cur.wait_stream(s2)
g2=capture(two)
print('fork2  %d nodes: host %.3f ms total %.3f ms'%((N,)+host_time(g2.replay)))
def many(k):
    def body():
        cur=torch.cuda.current_stream()
        for _ in range(k):
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1): work(N//(2*k),a)
            with torch.cuda.stream(s2): work(N//(2*k),b)
            cur.wait_stream(s1); cur.wait_stream(s2)
    return body
for k in (10,50):
    g=capture(many(k))
    print('fork2 x%d regions %d nodes: host %.3f ms total %.3f ms'%((k,N)+host_time(g.replay)))
# small linear graphs launched on two streams with events (segment graphs)
for k in (10,50):
    n=N//(2*k)
    ga=capture(lambda: work(n,a)); gb=capture(lambda: work(n,b))
    def run():
        cur=torch.cuda.current_stream()
        for _ in range(k):
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1): ga.replay()
            with torch.cuda.stream(s2): gb.replay()
            cur.wait_stream(s1); cur.wait_stream(s2)
    print('segment graphs x%d regions (%d graph launches, %d nodes): host %.3f ms total %.3f ms'%((k,2*k,N)+host_time(run)))
