"""wacv23_tsnet_amd -- MI355X-native forward path for TS-Net (WACV'23 motion retargeting).

Public surface mirrors the reference's model API (model/TSNet.py:203-407):
`TSNet(...)`, `set_test_input(...)`, `forward()` -> `rec_tar_img`; execution is a
C-ABI library of hand-written HIP kernels for gfx950 (see include/tsnet_abi.h).
"""
__all__ = ["prng"]
