"""Generate tests/golden/core_sim_api.json: class / method / argument NAMES of the reference's compiled module
`rcs._core.sim` and of the `rcs._core.common` base classes SimRobot / SimGripper inherit from, read from the reference's
own stub files (python/rcs/_core/sim.pyi, common.pyi).  Names only -- the contract a drop-in binding has to honour
(tests/test_host_logic.py::test_pybind_module_matches_the_reference_api).  Run where /root/reference exists.
"""
import ast
import json
import os

REF = "/root/reference/python/rcs/_core"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "core_sim_api.json")
WANT = {"sim.pyi": ["Sim", "SimConfig", "SimRobot", "SimRobotConfig", "SimRobotState", "SimGripper", "SimGripperConfig", "SimGripperState",
                    "SimCameraSet", "SimCameraConfig", "FrameSet", "CameraType"],
        "common.pyi": ["Robot", "Gripper", "BaseCameraConfig"]}


def main():
    api = {}
    for fname, classes in WANT.items():
        tree = ast.parse(open(os.path.join(REF, fname)).read())
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and node.name in classes:
                entry = {"bases": [ast.unparse(b).split(".")[-1] for b in node.bases], "methods": {}, "fields": []}
                for item in node.body:
                    if isinstance(item, ast.FunctionDef):
                        is_prop = any(isinstance(d, ast.Name) and d.id == "property" for d in item.decorator_list)
                        if is_prop:
                            entry["fields"].append(item.name)
                        else:
                            entry["methods"][item.name] = [a.arg for a in item.args.args if a.arg != "self"]
                    elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name):
                        entry["fields"].append(item.target.id)
                api[node.name] = entry
    json.dump(api, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, {k: len(v["methods"]) for k, v in api.items()})


if __name__ == "__main__":
    main()
