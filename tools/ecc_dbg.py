import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_ecc import _pair
from busca_amd import tracking
from oracle import ecc
for seed,(th,tx,ty) in enumerate([(0.01, 2.3, -1.4), (-0.02, -3.1, 0.8), (0.0, 0.4, 0.2)]):
    im1,im2,M=_pair(th=th,tx=tx,ty=ty,seed=seed)
    rho,Wo,trace=ecc.find_transform_ecc(ecc.bgr2gray(im1),ecc.bgr2gray(im2),motion="affine",return_trace=True)
    print("pair",seed,"oracle iters",len(trace))
    for k in (1,2,3,4,6,8,12,20,40,100):
        cc,W=tracking.find_transform_ecc(im1,im2,motion="MOTION_AFFINE",number_of_iterations=k,termination_eps=-1.0)
        ro,Wk=trace[min(k,len(trace))-1]
        print(" k=%3d gpu rho %.6f  oracle rho %.6f  |dW| %.2e" % (k,cc,ro,np.abs(W-Wk).max()))
