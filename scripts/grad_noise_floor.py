"""How far apart two fp32 CPU runs of the SAME network are: the oracle with and without mkldnn convs
(different summation order).  Deep-layer gradients differ by 1-2 % (relative L2) because rounding flips a
few ReLU masks per layer; this is the noise floor any fp32 implementation is compared against."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth, gast
from oracle.model import OracleDeeplabv2
from oracle.weights import det_state_dict
C=6
sd=det_state_dict("resnet50",C,False,seed=2333)
bc=synth.make_batch(B=2,H=256,W=256,C=C,k=2048,seed=2333)
def run(threads, mkldnn):
    torch.set_num_threads(threads)
    torch.backends.mkldnn.enabled = mkldnn
    om=OracleDeeplabv2(sd,"resnet50",C,False)
    ps1,ps2,fs=om(bc["images_s"])
    loss=gast.loss_calc([ps1,ps2],bc["label_s"])
    loss.backward()
    return {k:v.grad.clone() for k,v in om.named_parameters()}, ps1.detach()
a,pa=run(8,True); b,pb=run(8,False)
print("fwd diff", float((pa-pb).abs().max()/pa.abs().max()))
for k in ["encoder.resnet.conv1.weight","encoder.resnet.layer1.0.conv2.weight","encoder.resnet.layer3.2.conv3.weight","encoder.resnet.layer4.2.bn3.bias","encoder.resnet.layer4.2.conv3.weight","layer5.conv2d_list.0.weight"]:
    print(k, float((a[k]-b[k]).norm()/a[k].norm()))
