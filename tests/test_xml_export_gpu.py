"""GPU: the XML result export of the drop-in Tracker (`getScenarioElement`, `_storeTrackerArgs`, `_storeRun`; tracker.py:1469-1545,
pyTarget.py:745-829) against what the reference itself wrote for the same stream (tests/golden/g14_xml_export.npz: the serialised
<Tracker-settings> block and the <Track> elements of `_storeRun(preInitialized=False)`, terminated tracks included)."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_xml_export_matches_the_reference(gold_dir):
    from test_tracker_gpu import make_tracker
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    g = np.load(os.path.join(gold_dir, "g14_xml_export.npz"))
    sc = make_config("dense", seed=1234)
    trk, _ = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], sc["N"], 5.99, sc["x0"], sc["t0"], deviceTiming=True)      # (per-stage times: all nine <Runtime> records)
    for z, t in zip(sc["scans"], sc["times"]):
        trk.addMeasurementList(MeasurementList(float(t), z))
    scen = trk.getScenarioElement()
    trk._storeTrackerArgs(scen, name="dense", seed_of_stream=1234)
    trk._storeRun(scen, preInitialized=False, seed=7)
    assert scen.tag == "Scenario" and sorted("%s=%s" % kv for kv in scen.attrib.items()) == g["scenario_attrib"].tolist()
    assert ET.tostring(scen.find("Tracker-settings"), encoding="unicode") == str(g["settings"])
    run = scen.find("Run")
    assert sorted("%s=%s" % kv for kv in run.attrib.items()) == g["run_attrib"].tolist()
    rt = run.find("Runtime")
    assert rt.attrib == {"Description": "Per iteration", "precision": "6"}
    assert [e.tag for e in rt] == [s for s in g["runtime_stages"].tolist()]
    for e in rt:      # (wall-clock values: only the shape of the record)
        assert set(e.attrib) == {"mean", "min", "max"} and len(e.text.strip("[]").split()) == len(sc["scans"])
    tracks = [ET.tostring(e, encoding="unicode") for e in run.findall("Track")]
    assert tracks == g["tracks"].tolist()
    # a second run of the same scenario element counts up, and the full export writes every node of every chain
    trk._storeRun(scen, preInitialized=True)
    run2 = scen.findall("Run")[1]
    assert run2.attrib == {"i": "2"}
    for node, tr in zip(list(trk.getTrackNodes()), run2.findall("Track")):
        chain = node.backtrackNodes()
        assert tr.attrib["length"] == str(len(chain)) and len(tr.find("States")) == len(chain) and len(tr.find("SmoothedStates")) == 0
        first, last = tr.find("States")[0], tr.find("States")[-1]
        assert first.attrib["t"] == str(chain[0].time) and last.find("P").find("E").text == str(round(node.x_0[0], 2))
    trk.close()
