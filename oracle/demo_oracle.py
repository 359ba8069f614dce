"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the demo scripts' per-frame post-processing
(SURVEY.md section 8-f rank 2).  Only tests/ may import this module.

Follows demo/demo_face.py (identical lines in demo/demo_pose.py, given in brackets):
  * ref statistics      :180-182 [:186-188]  renorm_ref_img = ref_img / 255.0; .mean(dim=2), .std(dim=2) (unbiased)
  * frame re-normalise  :195-198 [:200-203]  (rec - gen_mean) / gen_std * ref_std + ref_mean
  * sample_img          :96-105  [:98-107]   CHW->HWC, + IMG_MEAN/255, clip [0,1], *255, cv2 BGR->RGB
  * uint8 conversion    :222     [:227]      Image.fromarray(rec_tar_img.astype('uint8'), "RGB")

Pin: every arithmetic step uses the very torch / numpy calls of the reference (same dtypes, same order).  The one
call that cannot run here is cv2.cvtColor(x, cv2.COLOR_BGR2RGB) (cv2 is not installed): on a float32 HxWx3 array it
is a pure channel reversal, restated as x[:, :, ::-1].  The demo scripts themselves cannot be imported (they parse
argv and run main() on hard-coded absolute paths at import) -- parity for this row is pinned to
the reference's operations, not to captured outputs of the script."""
import numpy as np
import torch

IMG_MEAN = np.array((101.84807705937696, 112.10832843463207, 111.65973036298041), dtype=np.float32)   # demo_face.py:27


def ref_statistics(ref_img: torch.Tensor):
    """ref_img: (1,3,H,W) float32, the first source image (mean-subtracted, 0..255 scale).  demo_face.py:180-182"""
    renorm_ref_img = ref_img / 255.0
    ref_mean = renorm_ref_img.view(1, 3, -1).mean(dim=2).view(1, 3, 1, 1)
    ref_std = renorm_ref_img.view(1, 3, -1).std(dim=2).view(1, 3, 1, 1)
    return ref_mean, ref_std


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


def postprocess_frame(rec_tar_imgs: torch.Tensor, ref_mean: torch.Tensor, ref_std: torch.Tensor) -> np.ndarray:
    """One generated frame (1,3,H,W) -> (H,W,3) uint8 RGB.  demo_face.py:195-199 and :222."""
    gen_mean = rec_tar_imgs.view(1, 3, -1).mean(dim=2).view(1, 3, 1, 1)
    gen_std = rec_tar_imgs.view(1, 3, -1).std(dim=2).view(1, 3, 1, 1)
    norm_rec_tar_imgs = (rec_tar_imgs - gen_mean) / gen_std
    rec_tar_imgs = norm_rec_tar_imgs * ref_std + ref_mean
    rec_tar_img = sample_img(rec_tar_imgs[0])
    return np.ascontiguousarray(rec_tar_img).astype('uint8')
