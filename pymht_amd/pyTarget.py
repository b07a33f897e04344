"""`Target`: one node of a track-hypothesis tree, as seen from the host.

Same constructor and field names as the reference's `pymht.pyTarget.Target` (pyTarget.py:16-40) so
scenario scripts, initiators and callers of `Tracker.getTrackNodes()` keep working.  In pymht_amd
the hypothesis forest itself lives in HBM (structure-of-arrays layers); `Target`
objects are what the host hands in (`Tracker.initiateTarget`) and what it gets back as *views* of
device nodes (`Tracker.getTrackNodes()`, `Tracker.__targetList__`): plain Python objects whose
`parent` / `trackHypotheses` links are materialised lazily from a snapshot of the device layers.
The XML result export (`_storeNode`, `_storeNodeSparse`) is here; plotting and pykalman smoothing of the reference class are out of scope.
"""
import copy
import datetime
import numpy as np

import xml.etree.ElementTree as ET

from .utils.xmlDefinitions import (activeTag, eastTag, idTag, inverseResidualCovarianceTag, lengthTag, mmsiTag, northTag, positionTag,
                                   smoothedstatesTag, statesTag, stateTag, timeTag, trackTag, velocityTag)


class Position:
    def __init__(self, *args, **kwargs):
        x, y = kwargs.get("x"), kwargs.get("y")
        if x is not None and y is not None:
            self.array = np.array([x, y])
        elif len(args) == 1:
            self.array = np.array(args[0])
        elif len(args) == 2:
            self.array = np.array([args[0], args[1]])
        else:
            raise ValueError("Invalid arguments to Position")

    def x(self):
        return self.array[0]

    def y(self):
        return self.array[1]

    def __str__(self):
        return "Pos: ({0: 8.2f},{1: 8.2f})".format(self.array[0], self.array[1])

    def __repr__(self):
        return "({0:.3e},{1:.3e})".format(self.array[0], self.array[1])


class Velocity(Position):
    def __str__(self):
        return "Vel: ({0: 5.2f},{1: 5.2f})".format(self.array[0], self.array[1])


class Target:
    _lazy_parent = None          # callable installed by the Tracker for device-backed views
    _lazy_children = None

    def __init__(self, time, scanNumber, x_0, P_0, ID=None, S_inv=None, **kwargs):
        assert (scanNumber is None) or (scanNumber == int(scanNumber))
        assert x_0.ndim == 1
        assert P_0.ndim == 2, str(P_0.shape)
        assert x_0.shape[0] == P_0.shape[0] == P_0.shape[1]
        self.isRoot = kwargs.get("isRoot", False)
        self.ID = ID
        self.time = time
        self.scanNumber = scanNumber
        self.x_0 = x_0
        self.P_0 = P_0
        self.S_inv = S_inv
        self.P_d = copy.copy(kwargs.get("P_d", 0.8))
        self._parent = kwargs.get("parent")
        self.measurementNumber = kwargs.get("measurementNumber", 0)
        self.measurement = kwargs.get("measurement")
        self.cumulativeNLLR = copy.copy(kwargs.get("cumulativeNLLR", 0))
        self._children = None
        self.mmsi = kwargs.get("mmsi")
        self.status = kwargs.get("status", activeTag)
        assert 0 <= self.P_d <= 1
        assert self._parent is None or isinstance(self._parent, Target)
        assert (self.mmsi is None) or (self.mmsi > 1e8)      # pyTarget.py:40

    # ---- tree links (lazy for device-backed views) ------------------------------------------
    @property
    def parent(self):
        if self._parent is None and self._lazy_parent is not None:
            self._parent = self._lazy_parent(self)
            self._lazy_parent = None
        return self._parent

    @parent.setter
    def parent(self, value):
        self._parent = value
        self._lazy_parent = None

    @property
    def trackHypotheses(self):
        if self._children is None and self._lazy_children is not None:
            self._children = self._lazy_children(self)
            self._lazy_children = None
        return self._children

    @trackHypotheses.setter
    def trackHypotheses(self, value):
        self._children = value
        self._lazy_children = None

    # ---- scalar queries (pyTarget.py:124-189) -----------------------------------------------
    def getScore(self):
        return self.cumulativeNLLR - self.getRoot().cumulativeNLLR

    def getRoot(self):
        node = self
        while node is not None and not node.isRoot:
            node = node.parent
        return node

    def getPosition(self):
        return Position(self.x_0[0:2])

    def getVelocity(self):
        return Velocity(self.x_0[2:4])

    def stepBack(self, stepsBack=1):
        node = self
        while stepsBack > 0 and node.parent is not None:
            node, stepsBack = node.parent, stepsBack - 1
        return node

    def getInitial(self):
        return self.stepBack(float("inf"))

    def getNumOfNodes(self):
        kids = self.trackHypotheses
        return 1 if kids is None else 1 + sum(k.getNumOfNodes() for k in kids)

    def depth(self, count=0):
        node = self
        while node.trackHypotheses is not None:
            node, count = node.trackHypotheses[0], count + 1
        return count

    def height(self, count=1):
        node = self
        while node.parent is not None:
            node, count = node.parent, count + 1
        return count

    def rootHeight(self, count=0):
        node = self
        while not (node.parent is None or node.isRoot):
            node, count = node.parent, count + 1
        return count

    def isOutsideRange(self, position, range):
        return np.linalg.norm(self.x_0[0:2] - position) > range

    def haveNoNeightbours(self, targetList, thresholdDistance):
        for target in targetList:
            for node in target.getLeafNodes():
                if np.linalg.norm(node.x_0[0:2] - self.x_0[0:2]) < thresholdDistance:
                    return False
        return True

    # ---- tree walks (pyTarget.py:414-471, :556-578) -----------------------------------------
    def getLeafNodes(self):
        out, stack = [], [self]
        while stack:
            node = stack.pop()
            kids = node.trackHypotheses
            if kids is None:
                out.append(node)
            else:
                stack.extend(reversed(kids))
        return out

    def getMeasurementSet(self, root=True):
        found, stack = set(), [(self, root)]
        while stack:
            node, top = stack.pop()
            if not top and node.measurementNumber not in (0, None):
                found.add((node.scanNumber, node.measurementNumber))
            for kid in (node.trackHypotheses or ()):
                stack.append((kid, False))
        return found

    def backtrackNodes(self, stepsBack=float("inf")):
        chain, node = [], self
        while node is not None:
            chain.append(node)
            node = node.parent
        return chain[::-1]

    # ---- XML result export (pyTarget.py:127-132, :297-302, :745-829): one <Track> per selected hypothesis -------------------------
    def getXmlStateStrings(self, precision=2):
        return tuple(str(round(self.x_0[i], precision)) for i in range(4))

    def _getHistoricalMmsi(self):
        node = self
        while node is not None:
            if getattr(node, "mmsi", None) is not None:
                return node.mmsi
            node = node.parent
        return None

    def _track_element(self, parent_element, attributes):
        """<Track id=.. [mmsi=..] ..> with an empty <States> child; returns (track, states)."""
        track = ET.SubElement(parent_element, trackTag)
        states = ET.SubElement(track, statesTag)
        mmsi = self._getHistoricalMmsi()
        if mmsi is not None:
            track.attrib[mmsiTag] = str(mmsi)
        track.attrib[idTag] = str(self.ID)
        for key, value in attributes.items():
            track.attrib[str(key)] = str(value)
        return track, states

    @staticmethod
    def _state_element(states, node):
        """<S t=..><P><N/><E/></P><V><N/><E/></V></S> for one node of the chain (north before east, as the reference writes it)."""
        east_p, north_p, east_v, north_v = node.getXmlStateStrings()
        el = ET.SubElement(states, stateTag, attrib={timeTag: str(node.time)})
        for tag, north, east in ((positionTag, north_p, east_p), (velocityTag, north_v, east_v)):
            pair = ET.SubElement(el, tag)
            ET.SubElement(pair, northTag).text = north
            ET.SubElement(pair, eastTag).text = east
        if node.status != activeTag:
            el.attrib[stateTag] = node.status
        return el

    def _storeNode(self, simulationElement, radarPeriod, **kwargs):
        """Every node of the chain root-of-time .. self (pyTarget.py:745-802).  The reference also writes a pykalman-smoothed copy
        (`getSmoothTrack`); smoothing is outside this package: the <SmoothedStates> element is there and empty, which is what the
        reference writes when its smoother reports failure."""
        track, states = self._track_element(simulationElement, kwargs)
        chain = self.backtrackNodes()
        track.attrib[lengthTag] = str(len(chain))
        ET.SubElement(track, smoothedstatesTag)
        for node in chain:
            el = self._state_element(states, node)
            if getattr(node, "S_inv", None) is not None:
                ET.SubElement(el, inverseResidualCovarianceTag).text = np.array_str(node.S_inv, max_line_width=9999)
        return track

    def _storeNodeSparse(self, simulationElement, **kwargs):
        """First and last node of the chain only (pyTarget.py:804-829)."""
        track, states = self._track_element(simulationElement, kwargs)
        chain = self.backtrackNodes()
        for node in ([chain[0], chain[-1]] if len(chain) > 1 else [chain[0]]):
            self._state_element(states, node)
        return track

    def backtrackPosition(self, stepsBack=float("inf")):
        return [n.x_0[0:2] for n in self.backtrackNodes()]

    def backtrackState(self, stepsBack=float("inf")):
        return [n.x_0 for n in self.backtrackNodes()]

    def backtrackMeasurement(self, stepsBack=float("inf")):
        return [n.measurement for n in self.backtrackNodes()]

    def __sub__(self, other):
        return self.x_0 - other.x_0

    def __repr__(self):
        stamp = datetime.datetime.fromtimestamp(self.time).strftime("%H:%M:%S.%f")
        out = "Time: " + stamp + "\t" + str(self.getPosition()) + " \t" + str(self.getVelocity())
        if self.ID is not None:
            out += " \tID: {:2}".format(self.ID)
        out += " \tcNLLR:" + "{: 06.4f}".format(float(self.cumulativeNLLR))
        if self.measurementNumber is not None and self.scanNumber is not None:
            out += " \tMeasurement(" + str(self.scanNumber) + ":" + str(self.measurementNumber) + ")"
        return out
