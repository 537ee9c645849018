import torch,sys
a=torch.load('gpurun_out/r06c51/t3.pt'); b=torch.load('gpurun_out/r06c51/t1.pt')
for i,n in ((0,256),(1,200)):
    dm=(a[i,:n,0]-b[i,:n,0]).abs(); 
    la=a[i,:n,0]+a[i,:n,1].log(); lb=b[i,:n,0]+b[i,:n,1].log()
    print(i,'max diff of max',float(dm.max()),'max |logit max|',float(a[i,:n,0].abs().max()),'log Z diff max',float((la-lb).abs().max()), 'min/max sum', float(b[i,:n,1].min()), float(b[i,:n,1].max()))
