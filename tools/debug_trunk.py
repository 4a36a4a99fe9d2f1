import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
from seeded_init import seeded_state_dict
from hifihr_amd import ops
from hifihr_amd.network import ResNet18Trunk
g = np.load(os.path.join(R, "tests/golden/resnet18_small.npz"))
net = ResNet18Trunk(layer4_stride=1, conv_impl="mfma")
net.load_state_dict(seeded_state_dict(net))
net = net.cuda().train()
x = ops.image_to_nhwc4(torch.tensor(g["x"]).cuda())
def S(tag):
    torch.cuda.synchronize(); print(tag, flush=True)
h = net.maxpool(net.relu(net.bn1(net.conv1(x)))); S("stem")
low = net.layer2(net.layer1(h)); S("l2")
feat = net.layer4(net.layer3(low)); S("l4")
loss = (low * torch.tensor(g["wl"]).cuda()).sum() + (feat * torch.tensor(g["wf"]).cuda()).sum(); S("loss")
# backward piecewise with hooks
for name, m in net.named_modules():
    if type(m).__name__ == "Conv2dMFMA":
        m.register_full_backward_hook(lambda mod, gi, go, name=name: (torch.cuda.synchronize(), print("bwd done", name, flush=True)))
loss.backward(); S("bwd")
