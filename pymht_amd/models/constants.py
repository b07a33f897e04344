"""Model constants of the CV ("pv") tracker model.  Mirrors the names of the reference's
pymht/models/constants.py:2-10 so scenario scripts written against pyMHT keep working."""
import numpy as np

defaultType = np.float32        # all model matrices are f32 (load-bearing: SURVEY.md fact 4)
nDimState = 4
nObsDim_AIS = 4
sigmaR_RADAR_tracker = 2.5      # measurement std-dev used by the filter [m]
sigmaR_RADAR_true = 2.5
sigmaQ_tracker = 1.0
sigmaQ_true = 1.0
