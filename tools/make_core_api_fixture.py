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
# rcs._core.common itself (src/pybind/rcs.cpp:224-336): the classes SURVEY 8b lists as "classes a replacement must expose",
# plus the module-level functions and exported enum constants -> tests/golden/core_common_api.json
COMMON = ["Pose", "RPY", "Kinematics", "Pin", "RobotType", "RobotPlatform", "RobotMetaConfig", "RobotConfig", "BaseCameraConfig", "GraspType"]
COMMON_FUNCTIONS = ["robots_meta_config", "FrankaHandTCPOffset", "IdentityRotMatrix", "IdentityRotQuatVec", "IdentityTranslation"]
OUT_COMMON = os.path.join(os.path.dirname(OUT), "core_common_api.json")


def class_entry(node):
    """Names of one stub class: bases, methods -> argument names (every overload of an @typing.overload family), fields."""
    entry = {"bases": [ast.unparse(b).split(".")[-1] for b in node.bases], "methods": {}, "overloads": {}, "fields": []}
    for item in node.body:
        if isinstance(item, ast.FunctionDef):
            is_prop = any(isinstance(d, ast.Name) and d.id == "property" for d in item.decorator_list)
            args = [a.arg for a in item.args.args if a.arg != "self"]
            if is_prop:
                entry["fields"].append(item.name)
            elif any("overload" in ast.unparse(d) for d in item.decorator_list):
                entry["overloads"].setdefault(item.name, []).append(args)
                entry["methods"].setdefault(item.name, [])
            else:
                entry["methods"][item.name] = args
        elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name):
            entry["fields"].append(item.target.id)
    return entry


def main():
    api = {}
    for fname, classes in WANT.items():
        tree = ast.parse(open(os.path.join(REF, fname)).read())
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and node.name in classes:
                entry = class_entry(node)
                entry.pop("overloads")
                api[node.name] = entry
    json.dump(api, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, {k: len(v["methods"]) for k, v in api.items()})
    tree = ast.parse(open(os.path.join(REF, "common.pyi")).read())
    common = {"classes": {}, "functions": {}, "constants": []}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in COMMON:
            common["classes"][node.name] = class_entry(node)
        elif isinstance(node, ast.FunctionDef) and node.name in COMMON_FUNCTIONS:
            common["functions"][node.name] = [a.arg for a in node.args.args]
        elif isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name) and ast.unparse(node.annotation) in ("RobotType", "RobotPlatform", "GraspType"):
            common["constants"].append(node.target.id)
    # the RL IK class lives in an extension module of its own (extensions/rcs_robotics_library: rcs_robotics_library._core.rl)
    rl_stub = "/root/reference/extensions/rcs_robotics_library/src/rcs_robotics_library/_core/rl.pyi"
    common["rl"] = {}
    for node in ast.parse(open(rl_stub).read()).body:
        if isinstance(node, ast.ClassDef) and node.name == "RoboticsLibraryIK":
            common["rl"][node.name] = class_entry(node)
    json.dump(common, open(OUT_COMMON, "w"), indent=1, sort_keys=True)
    print("wrote", OUT_COMMON, {k: len(v["methods"]) for k, v in common["classes"].items()}, common["functions"], common["constants"])


if __name__ == "__main__":
    main()
