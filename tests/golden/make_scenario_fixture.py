#!/usr/bin/env python3
"""
Cuts tests/golden/scenarios/USA_Lanker-2_18_T-1_route.xml out of the reference's scenario file: the lanelets the route of
planning problem 21007 touches plus their neighbours (predecessors, successors, adjacent lanes, the other goal lanelets) and the
planning problem itself -- DATA of a CommonRoad scenario -- and, of its 83 dynamic obstacles (which the lane-following use case
never reads), the four that come closest to the recorded ego trajectory (for the collision verdict of row f4).  Run in the build container: `python tests/golden/make_scenario_fixture.py`.
"""
import os
import xml.etree.ElementTree as ET

SRC = "/root/reference/scenarios/USA_Lanker-2_18_T-1.xml"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenarios", "USA_Lanker-2_18_T-1_route.xml")
ROUTE = [3672, 3452, 3454, 3456]
OBSTACLES = [2829, 2839, 2716, 2858]


def main():
    root = ET.parse(SRC).getroot()
    lan = {int(l.attrib["id"]): l for l in root.findall("lanelet")}
    keep = set(ROUTE)
    pp = root.find("planningProblem")
    keep |= {int(g.attrib["ref"]) for g in pp.findall("goalState/position/lanelet")}
    for lid in list(keep):
        for tag in ("predecessor", "successor", "adjacentLeft", "adjacentRight"):
            keep |= {int(n.attrib["ref"]) for n in lan[lid].findall(tag)}
    keep = {k for k in keep if k in lan}
    out = ET.Element("commonRoad", root.attrib)
    for lid in sorted(keep):
        out.append(lan[lid])
    for o in root.findall("obstacle"):
        if int(o.attrib["id"]) in OBSTACLES:
            out.append(o)
    out.append(pp)
    ET.indent(out) if hasattr(ET, "indent") else None
    ET.ElementTree(out).write(OUT, xml_declaration=True, encoding="utf-8")
    print("wrote", OUT, "with lanelets", sorted(keep), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
