"""Diagnostic (GPU box): a decoder conv_block with the synthetic weights at small spatial sizes, sparse upstream gradient."""
import sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from nextbestpath_amd.networks import training as tr
from nextbestpath_amd.networks.nbp_model import NBP
from nextbestpath_amd.utility.synthetic import make_nbp_state_dict
D = "cuda"
sd = make_nbp_state_dict(9)
for HW in (16, 8, 32):
    torch.manual_seed(HW)
    net = NBP(); net.load_state_dict(sd); net.train()
    blk = net.Up_conv4_1.conv
    B, C = 4, 256
    a, dd = torch.relu(torch.randn(B, C, HW, HW)), torch.relu(torch.randn(B, C, HW, HW))
    gy = torch.zeros(B, 256, HW, HW)
    idx = (torch.randint(0, B, (40,)), torch.randint(0, 256, (40,)), torch.randint(0, HW, (40,)), torch.randint(0, HW, (40,)))
    gy.index_put_(idx, torch.randn(40))
    def ref(dt):
        m = NBP(); m.load_state_dict(sd); m = m.to(dt).train()
        bb = m.Up_conv4_1.conv
        aa, d2 = a.to(dt).clone().requires_grad_(True), dd.to(dt).clone().requires_grad_(True)
        bb(torch.cat((aa, d2), 1)).backward(gy.to(dt))
        return [bb[0].weight.grad.double(), bb[3].weight.grad.double(), aa.grad.double(), d2.grad.double(), bb[1].weight.grad.double()]
    r64, r32 = ref(torch.float64), ref(torch.float32)
    netd = NBP(); netd.load_state_dict(sd); netd = netd.to(D).train()
    bd = netd.Up_conv4_1.conv
    ad = a.permute(0, 2, 3, 1).contiguous().to(D).requires_grad_(True); ddd = dd.permute(0, 2, 3, 1).contiguous().to(D).requires_grad_(True)
    tr._block(bd, ad, ddd).backward(gy.permute(0, 2, 3, 1).contiguous().to(D))
    hip = [bd[0].weight.grad.cpu().double(), bd[3].weight.grad.cpu().double(), ad.grad.permute(0, 3, 1, 2).cpu().double(),
           ddd.grad.permute(0, 3, 1, 2).cpu().double(), bd[1].weight.grad.cpu().double()]
    for nm, h, t, c in zip(("dW conv.0", "dW conv.3", "d a", "d dd", "dgamma.1"), hip, r32, r64):
        n = float(c.norm()); print(f"HW={HW:2d} {nm:10s} hip {float((h-c).norm())/n:.2e}  torch32 {float((t-c).norm())/n:.2e}")
