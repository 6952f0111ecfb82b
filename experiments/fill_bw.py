

def main():
    import torch, time
    dev='cuda'
    for gb in (4, 16, 64):
        n=int(gb*1e9/4)
        x=torch.empty(n,device=dev,dtype=torch.float32)
        for _ in range(2): x.fill_(1.0)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): x.fill_(2.0)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/5
        print('fill %d GB: %.2f ms = %.2f TB/s'%(gb,ms,gb/ms))
        y=torch.empty(n//2,device=dev,dtype=torch.float32)
        e0.record()
        for _ in range(5): y.copy_(x[:n//2])
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/5
        print('copy %d GB read + %d GB write: %.2f ms = %.2f TB/s total'%(gb//2,gb//2,ms,gb/ms))
        del x,y


if __name__ == "__main__":
    main()
