"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the demo scripts' per-frame post-processing
(SURVEY.md section 8-f rank 2).  Only tests/ may import this module.

Follows demo/demo_face.py (identical lines in demo/demo_pose.py, given in brackets):
  * ref statistics      :180-182 [:186-188]  renorm_ref_img = ref_img / 255.0; .mean(dim=2), .std(dim=2) (unbiased)
  * frame re-normalise  :195-198 [:200-203]  (rec - gen_mean) / gen_std * ref_std + ref_mean
  * sample_img          :96-105  [:98-107]   CHW->HWC, + IMG_MEAN/255, clip [0,1], *255, cv2 BGR->RGB
  * uint8 conversion    :222     [:227]      Image.fromarray(rec_tar_img.astype('uint8'), "RGB")

Pin: tests/golden/g8_demo_post.npz holds the bytes the reference's OWN statements produce on PRNG frames --
oracle/capture_demo_goldens.py lifts them out of demo/demo_face.py with `ast` and executes them (the script itself cannot be
imported: it parses argv and runs main() on absolute paths at import) -- and tests/test_demo_post.py checks that this restatement
reproduces them bit for bit.  The one call that cannot run here is cv2.cvtColor(x, cv2.COLOR_BGR2RGB) (cv2 is not installed): on a
float32 HxWx3 array it is a pure channel reversal, x[:, :, ::-1], in the capture and here alike.

`stats64=True` evaluates the four statistics in fp64 (then rounds them to fp32, the dtype every later operation keeps): torch's fp32
mean / std carry the summation-order noise of whatever vectorised reduction the host picks, a well-defined fp64 statistic does not --
that is the variant the device bytes are required to EQUAL."""
import numpy as np
import torch

IMG_MEAN = np.array((101.84807705937696, 112.10832843463207, 111.65973036298041), dtype=np.float32)   # demo_face.py:27


def _mean_std(x: torch.Tensor, stats64: bool):
    v = x.view(1, 3, -1)
    if stats64:
        return v.double().mean(dim=2).float().view(1, 3, 1, 1), v.double().std(dim=2).float().view(1, 3, 1, 1)
    return v.mean(dim=2).view(1, 3, 1, 1), v.std(dim=2).view(1, 3, 1, 1)


def ref_statistics(ref_img: torch.Tensor, stats64: bool = False):
    """ref_img: (1,3,H,W) float32, the first source image (mean-subtracted, 0..255 scale).  demo_face.py:180-182"""
    renorm_ref_img = ref_img / 255.0
    return _mean_std(renorm_ref_img, stats64)


def sample_img(rec_img_batch: torch.Tensor) -> np.ndarray:
    """demo_face.py:96-105 (rec_img_batch: (3,H,W))."""
    rec_img = rec_img_batch.data.cpu().numpy()
    img_mean = (torch.from_numpy(IMG_MEAN) / 255).data.cpu().numpy()
    rec_img = rec_img.transpose(1, 2, 0)
    rec_img = rec_img + img_mean
    rec_img[rec_img < 0] = 0
    rec_img[rec_img > 1] = 1
    rec_img *= 255
    rec_img = rec_img[:, :, ::-1]            # cv2.cvtColor(rec_img, cv2.COLOR_BGR2RGB)
    return rec_img


def postprocess_frame(rec_tar_imgs: torch.Tensor, ref_mean: torch.Tensor, ref_std: torch.Tensor, stats64: bool = False) -> np.ndarray:
    """One generated frame (1,3,H,W) -> (H,W,3) uint8 RGB.  demo_face.py:195-199 and :222."""
    gen_mean, gen_std = _mean_std(rec_tar_imgs, stats64)
    norm_rec_tar_imgs = (rec_tar_imgs - gen_mean) / gen_std
    rec_tar_imgs = norm_rec_tar_imgs * ref_std + ref_mean
    rec_tar_img = sample_img(rec_tar_imgs[0])
    return np.ascontiguousarray(rec_tar_img).astype('uint8')
