"""Input containers of the scan path.  Only what `Tracker.addMeasurementList` consumes:
`MeasurementList` (reference pymht/utils/classDefinitions.py:560-594) and the default-argument
`AisMessageList` (:597-626).  Simulation targets, plotting and XML are out of scope."""
import datetime
import numpy as np


class MeasurementList:
    """One radar scan: `time` [s] and `measurements`, an (M,2) ndarray (f32 from the simulator)."""

    def __init__(self, time, measurements=None):
        self.time = time
        self.measurements = measurements if measurements is not None else []

    def __eq__(self, other):
        return self.time == other.time and np.array_equal(self.measurements, other.measurements)

    def __str__(self):
        stamp = datetime.datetime.fromtimestamp(self.time).strftime("%H:%M:%S.%f")
        return "Time: " + stamp + "\tMeasurements:\t" + ", ".join(str(m) for m in self.measurements)

    __repr__ = __str__

    def filterUnused(self, unused_measurement_indices):
        return MeasurementList(self.time, self.measurements[np.where(unused_measurement_indices)])

    def getTimeString(self, timeFormat="%H:%M:%S"):
        return datetime.datetime.fromtimestamp(self.time).strftime(timeFormat)

    def getMeasurements(self):
        return self.measurements


from ..ais import AisMessage, AisMessageList      # noqa: E402,F401  (classDefinitions.py:428-434, :597-622)
AIS_message = AisMessage                            # (the reference's name)
