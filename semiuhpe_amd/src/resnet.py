"""ResNet-50 backbone with torchvision's module tree (reference src/networks.py:443-500 builds `models.resnet50()` and replaces
`fc` by the regression head; released checkpoint DAD-COCOHead-ResNet50-best.pth).  Bottleneck v1.5 (stride on
the 3x3 conv).  Layers with explicit forward/backward over the HIP kernels; every BatchNorm's statistics come out of the
producing conv's epilogue, the residual add and the ReLU ride in the last BatchNorm's apply kernel.
"""
from .. import kernels as K
from ..layers import BatchNorm2d, Conv2d, GlobalAvgPool, Layer, MaxPool2d, Network, Sequential, conv_bn_fwd
from .efficientnet_v2 import init_regression_head_, regression_head


class Bottleneck(Layer):
    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = Conv2d(cin, planes, 1)
        self.bn1 = BatchNorm2d(planes, act="relu")
        self.conv2 = Conv2d(planes, planes, 3, stride, 1)
        self.bn2 = BatchNorm2d(planes, act="relu")
        self.conv3 = Conv2d(planes, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4, act="relu")          # relu(bn3(conv3) + identity): the block's last ReLU
        self.downsample = downsample
        self.need_input_grad = True

    def fwd(self, x, tape):
        out = conv_bn_fwd(self.conv1, self.bn1, x, tape)
        out = conv_bn_fwd(self.conv2, self.bn2, out, tape)
        identity = x if self.downsample is None else conv_bn_fwd(self.downsample[0], self.downsample[1], x, tape)
        return conv_bn_fwd(self.conv3, self.bn3, out, tape, residual=identity)

    def bwd(self, dout, tape):
        out = tape[-1][4] if tape[-1][4] is not None else None       # forward output kept by bn3 for the ReLU mask
        d = self.conv3.bwd(self.bn3.bwd(dout, tape), tape)
        g = K.act_bwd(out, dout, out.shape[-1], "relu")              # gradient entering the identity path
        if self.downsample is not None:
            d_id = self.downsample[0].bwd(self.downsample[1].bwd(g, tape), tape)
        else:
            d_id = g
        from .. import layers as L
        fuse = L.SKIP_ADD_FUSION and self.conv1.need_input_grad       # the identity path's gradient rides in conv1's input-gradient epilogue
        d = self.conv1.bwd(self.bn1.bwd(self.conv2.bwd(self.bn2.bwd(d, tape), tape), tape), tape, skip=d_id if fuse else None)
        return d if fuse else K.add(d, d_id)


class ResNet50(Network):
    def __init__(self, num_classes=9):
        super().__init__()
        self.conv1 = Conv2d(3, 64, 7, 2, 3)
        self.conv1.need_input_grad = False
        self.bn1 = BatchNorm2d(64, act="relu")
        self.maxpool = MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            layers = []
            for b in range(blocks):
                s = stride if b == 0 else 1
                down = None
                if s != 1 or cin != planes * 4:
                    down = Sequential(Conv2d(cin, planes * 4, 1, s), BatchNorm2d(planes * 4))
                layers.append(Bottleneck(cin, planes, s, down))
                cin = planes * 4
            setattr(self, f"layer{i}", Sequential(*layers))
        self.avgpool = GlobalAvgPool()
        self.fc = regression_head(2048, 512, 128, num_classes)
        self.init_like_reference_()
        init_regression_head_(self.fc)

    def _stages(self):
        return [self.layer1, self.layer2, self.layer3, self.layer4]

    def fwd(self, x, tape):
        x = self.maxpool.fwd(conv_bn_fwd(self.conv1, self.bn1, x, tape), tape)
        for s in self._stages():
            x = s.fwd(x, tape)
        return self.fc.fwd(self.avgpool.fwd(x, tape), tape)

    def bwd(self, dy, tape):
        dy = self.avgpool.bwd(self.fc.bwd(dy, tape), tape)
        for s in reversed(self._stages()):
            dy = s.bwd(dy, tape)
        dy = self.bn1.bwd(self.maxpool.bwd(dy, tape), tape)
        self.conv1.bwd(dy, tape)
