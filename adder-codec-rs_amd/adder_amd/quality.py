"""Quality metrics of a reconstruction against its source frame -- the host-side mirror of
adder-codec-rs/src/utils/cv.rs:306-360 (`calculate_quality_metrics`: MSE over every element in f64, an MSE of exactly 0
replaced by 1e-7 "so that PSNR isn't undefined", PSNR = 20 log10(255) - 10 log10(MSE)).  The reference applies it to the input
frame against the transcoder's running intensities (framed.rs:136-151, feature "feature-logging") and, in its viewer, against
the framer's reconstruction (adder-viz/src/transcoder/adder.rs:318): the harness of SURVEY 8(f)1 does the latter.  The squared
differences are integers below 2^16 and their sum stays far below 2^53, so the f64 sum is exact in any order: numpy's equals
the reference's sequential loop bit for bit.  (SSIM -- cv.rs:362-430 -- is not mirrored.)"""
import math

import numpy as np


def calculate_mse(original, reconstructed):
    a = np.asarray(original)
    b = np.asarray(reconstructed)
    if a.shape != b.shape:
        raise ValueError("Shapes of original and reconstructed images must match")   # cv.rs:312, :336
    d = a.astype(np.int64) - b.astype(np.int64)
    return float(int((d * d).sum())) / float(a.size)


def calculate_psnr(mse):
    return 20.0 * math.log10(255.0) - 10.0 * math.log10(mse)


def calculate_quality_metrics(original, reconstructed):
    """-> {"mse": .., "psnr": ..} exactly as cv.rs:306-333 fills QualityMetrics {mse: Some, psnr: Some, ssim: None}."""
    mse = calculate_mse(original, reconstructed)
    if mse == 0.0:
        mse = 0.0000001
    return {"mse": mse, "psnr": calculate_psnr(mse)}
