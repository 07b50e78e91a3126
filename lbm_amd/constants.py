"""Script-level constants of the reference, kept by name and default value.

training/preprocess.py:16-36, multi-train/withpsd.py:15-66, training/model.py:31-54.
"""
FILTER_LFREQ = 4.0
FILTER_HFREQ = 30.0
NOTCH_FREQ = 60.0
ICA_N_COMPONENTS = None
ICA_RANDOM_STATE = 97
ICA_METHOD = "fastica"
AUTO_DETECT_EOG = True
DEVIATION_CODES = [251, 252]   # training/preprocess.py:26
RESPONSE_CODE = 253            # :27
RT_FILTER_THRESHOLD = 0.1      # :31
TMIN = -0.5
TMAX = 1.5
BASELINE = (None, 0)
REJECT_CRITERIA = dict(eeg=150e-6)
PSD_METHOD = "welch"
PSD_FMIN = FILTER_LFREQ
PSD_FMAX = FILTER_HFREQ
PSD_WINDOW = "hann"
VQ_EMBEDDING_DIM = 256
VQ_NUM_EMBEDDINGS = 512
VQ_COMMITMENT_COST = 0.25
VQ_DECAY = 0.99
BATCH_SIZE = 64
ZSCORE_EPSILON = 1e-6  # training/preprocess.py:284
