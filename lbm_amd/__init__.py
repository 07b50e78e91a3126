"""lbm_amd -- MI355X (gfx950) native batched EEG epoch pipeline: FIR band-pass, ICA apply, epoch
post-processing, Welch PSD + band powers, VQ-VAE encoder + nearest-codebook tokenisation.

Host code mirrors the call surface of the reference scripts (beetleberries/LBM: training/preprocess.py,
multi-train/withpsd.py, preprocessing_unsupervised.py, training/model.py); the arithmetic runs in
hand-written HIP kernels behind the C ABI of include/lbm_abi.h (lbm_amd/liblbm_hip.so).  No CPU fallback.
"""
from .constants import *  # noqa: F401,F403
from . import _lib  # noqa: F401
from .filter import create_filter, filter_data, notch_filter, fir_apply, resample  # noqa: F401
from .ica import ica_apply, ica_get_sources, ica_projection, affine_channels, average_reference_matrix  # noqa: F401
from .fastica import ica_fit, find_bads_eog, find_bads_ecg, fastica_parallel, ica_stats, channel_moments  # noqa: F401
from .epochs import (epoch_post, affine_post, affine_post_supported, standardize_epochs, Scaler, n_baseline_samples, make_epochs,  # noqa: F401
                     save_preprocess_outputs, save_withpsd_outputs)
from .psd import psd_array_welch, band_powers, classify_eeg_state, classify_eeg_state_variant  # noqa: F401
from .quantizer import (VectorQuantizerEMA, VQVAE, encode_data_for_transformer, vq_argmin, vq_split_tables, fold_projection, dedup_codebook, remap_codes, conv_split_tables, conv_wino3_tables, conv1d,  # noqa: F401
                        encoder_forward, rownorm2, vq_gather_loss, perplexity_from_hist, vq_ema_sums, vq_ema_apply)
from .recording import (read_raw_eeglab, pick_eeg_channels, extract_events, pair_trials, vigilance_labels,  # noqa: F401
                        mne_events, preprocess_recording, preprocess_recording_withpsd,
                        process_unsupervised_recording)
from .dataset import (classify_channels, preprocess_dataset_withpsd, process_unsupervised_dataset, withpsd_file,  # noqa: F401
                      unsupervised_file, SkipFile)
from .pipeline import EpochPipeline, PipelineConfig, shard_range, gather_epoch_outputs, gather_pipeline_outputs, allreduce_hist  # noqa: F401
