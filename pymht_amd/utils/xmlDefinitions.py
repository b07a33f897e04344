"""Tag strings of `Target.status` and of the XML result export (`Tracker._storeRun`, `Target._storeNode*`): the element and
attribute names ARE the file format the reference's evaluation scripts read (reference pymht/utils/xmlDefinitions.py)."""
# track status (pymht/utils/xmlDefinitions.py:37-41)
preinitializedTag = "preinitialized"
activeTag = "Active"
outofrangeTag = "OutOfRange"
toolowscoreTag = "TooLowScore"
# elements
scenarioTag, trackerSettingsTag, runTag, runtimeTag = "Scenario", "Tracker-settings", "Run", "Runtime"
trackTag, statesTag, smoothedstatesTag, stateTag = "Track", "States", "SmoothedStates", "S"
positionTag, velocityTag, northTag, eastTag = "P", "V", "N", "E"
inverseResidualCovarianceTag = "S_inv"
# attributes
mmsiTag, timeTag, idTag, iterationTag, seedTag, lengthTag = "mmsi", "t", "id", "i", "seed", "length"
meanTag, minTag, maxTag, precisionTag, descriptionTag, terminatedTag = "mean", "min", "max", "precision", "Description", "terminated"
timeLogPrecision = 6
