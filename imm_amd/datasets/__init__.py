"""Input pipelines of the reference (imm/datasets/*) on the MI355X data path: JPEG decode on host threads, everything
after the decoder (to-float, bilinear resize, central crop, mask, thin-plate-spline pair) on the GPU."""
from .celeba_dataset import CelebADataset       # noqa: F401
from .aflw_dataset import AFLWDataset           # noqa: F401
from .tps_dataset import TPSDataset             # noqa: F401
from .impair_dataset import ImagePairDataset    # noqa: F401
