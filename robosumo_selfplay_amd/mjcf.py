"""MJCF-subset scene compiler for the RoboSumo two-agent scenes.

Replaces, for this hot path only, the reference's one-time model build:

* scene assembly   -- reference ``robosumo/robosumo/envs/utils.py:46-183`` (``construct_scene``):
  world file + two agent files, per-agent default class with density override, tatami / border
  resize, initial placement on a circle of radius 1.5, ``<scope>/`` name prefixes.
* model compile    -- what ``mj_loadXML`` does for that scene (``mujoco-py/mujoco_py/cymj.pyx:162-176``):
  defaults resolution, ``fromto`` geoms, inertia-from-geom, joint/dof tables, weld ids for the
  collision filter, ``qpos0``, and the ``mj_setConst`` constants (``dof_invweight0``,
  ``body_invweight0``, ``meaninertia``) evaluated at ``qpos0``.
* agent index bookkeeping -- reference ``agents.py:45-115`` (qpos/qvel ranges, body ids, action
  bounds from ``ctrlrange``).

Only the MJCF elements that the reference assets use are understood (compiler, option, default
[one nested class level], worldbody/body/geom/joint, actuator/motor); anything else that could
change the physics raises ``MjcfError`` instead of being silently dropped.

The result is a ``SumoModel`` (flat numpy tables).  ``SumoModel.to_blob()`` serialises it in the
format declared in ``include/sumo_model.h`` which both the HIP engine and the CPU oracle read.
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX = 0, 2, 3, 5, 6  # mjtGeom enum values
JNT_FREE, JNT_HINGE = 0, 3                                                      # mjtJoint enum values
_GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
               "cylinder": GEOM_CYLINDER, "box": GEOM_BOX}
MJ_MINVAL = 1e-15


class MjcfError(ValueError):
    pass


# ----------------------------------------------------------------------------------------------
# small math helpers (numpy, float64)
# ----------------------------------------------------------------------------------------------
def _floats(s: str) -> np.ndarray:
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat2mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def quat_z2vec(vec):
    """Quaternion rotating the z axis onto unit vector ``vec`` (MuJoCo ``mjuu_z2quat``)."""
    axis = np.cross([0.0, 0.0, 1.0], vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        axis = np.array([1.0, 0.0, 0.0])
    else:
        axis = axis / s
    ang = math.atan2(s, vec[2])
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


# ----------------------------------------------------------------------------------------------
# intermediate tree
# ----------------------------------------------------------------------------------------------
@dataclass
class _Geom:
    name: str
    type: int
    size: np.ndarray            # (3,) mujoco convention
    pos: np.ndarray
    quat: np.ndarray
    density: float
    friction: np.ndarray
    margin: float
    gap: float
    condim: int
    contype: int
    conaffinity: int
    solref: np.ndarray
    solimp: np.ndarray
    solmix: float
    priority: int


@dataclass
class _Joint:
    name: str
    type: int
    pos: np.ndarray
    axis: np.ndarray
    limited: bool
    range: np.ndarray           # radians for hinges
    armature: float
    damping: float
    stiffness: float
    margin: float
    ref: float


@dataclass
class _Body:
    name: str
    pos: np.ndarray
    quat: np.ndarray
    geoms: List[_Geom] = field(default_factory=list)
    joints: List[_Joint] = field(default_factory=list)
    children: List["_Body"] = field(default_factory=list)


@dataclass
class _Motor:
    name: str
    joint: str
    gear: float
    ctrllimited: bool
    ctrlrange: np.ndarray


_GEOM_BUILTIN = dict(type="sphere", size="0 0 0", pos="0 0 0", density="1000", friction="1 0.005 0.0001",
                     margin="0", gap="0", condim="3", contype="1", conaffinity="1",
                     solref="0.02 1", solimp="0.9 0.95 0.001 0.5 2", solmix="1", priority="0")
_JOINT_BUILTIN = dict(type="hinge", pos="0 0 0", axis="0 0 1", limited="false", range="0 0", armature="0",
                      damping="0", stiffness="0", margin="0", ref="0", frictionloss="0")
_MOTOR_BUILTIN = dict(gear="1", ctrllimited="false", ctrlrange="0 0")
# attributes that do not influence the dynamics and may be ignored
_COSMETIC = {"rgba", "material", "name", "class"}


def _merged(builtin: Dict[str, str], *layers: Optional[Dict[str, str]]) -> Dict[str, str]:
    out = dict(builtin)
    for layer in layers:
        if layer:
            out.update(layer)
    return out


def _bool(s: str) -> bool:
    if s not in ("true", "false"):
        raise MjcfError("bad boolean %r" % s)
    return s == "true"


class _Defaults:
    """Main default + one level of named child classes (all the reference scenes need,
    ``utils.py:118-144``).  A child class inherits every element of the main default."""

    def __init__(self):
        self.main: Dict[str, Dict[str, str]] = {"geom": {}, "joint": {}, "motor": {}}
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {}

    def load_main(self, elem: Optional[ET.Element]):
        if elem is None:
            return
        for child in elem:
            if child.tag == "default":
                raise MjcfError("nested defaults in the world file are not supported")
            if child.tag not in self.main:
                raise MjcfError("unsupported default element <%s>" % child.tag)
            self.main[child.tag].update(child.attrib)

    def add_class(self, name: str, elems: Sequence[ET.Element], geom_override: Dict[str, str]):
        cls = {"geom": {}, "joint": {}, "motor": {}}
        seen_geom = False
        for child in elems:
            if child.tag not in cls:
                raise MjcfError("unsupported default element <%s>" % child.tag)
            cls[child.tag].update(child.attrib)
            if child.tag == "geom":
                cls["geom"].update(geom_override)
                seen_geom = True
        if not seen_geom:
            # utils.py:135-144: a fresh geom default with density/contype/conaffinity
            cls["geom"].update({"contype": "1", "conaffinity": "1"})
            cls["geom"].update(geom_override)
        self.classes[name] = cls

    def resolve(self, tag: str, builtin: Dict[str, str], attrib: Dict[str, str]) -> Dict[str, str]:
        cls = attrib.get("class")
        layers = [self.main[tag]]
        if cls is not None:
            if cls not in self.classes:
                raise MjcfError("unknown default class %r" % cls)
            layers.append(self.classes[cls][tag])
        return _merged(builtin, *layers, attrib)


def _check_known(tag: str, attrib: Dict[str, str], known: Sequence[str]):
    for k in attrib:
        if k not in known and k not in _COSMETIC:
            raise MjcfError("unsupported attribute %s=%r on <%s>" % (k, attrib[k], tag))


def _make_geom(elem: ET.Element, defaults: _Defaults, angle_scale: float) -> _Geom:
    a = defaults.resolve("geom", _GEOM_BUILTIN, dict(elem.attrib))
    _check_known("geom", a, list(_GEOM_BUILTIN) + ["fromto"])
    if a["type"] not in _GEOM_TYPES:
        raise MjcfError("unsupported geom type %r" % a["type"])
    gtype = _GEOM_TYPES[a["type"]]
    size_in = _floats(a["size"])
    size = np.zeros(3)
    pos = _floats(a["pos"])
    quat = np.array([1.0, 0, 0, 0])
    if "fromto" in a:
        if gtype not in (GEOM_CAPSULE, GEOM_CYLINDER):
            raise MjcfError("fromto only supported on capsule/cylinder")
        ft = _floats(a["fromto"])
        vec = ft[0:3] - ft[3:6]          # MuJoCo: z axis points from the second to the first point
        length = np.linalg.norm(vec)
        if length < MJ_MINVAL:
            raise MjcfError("degenerate fromto")
        pos = 0.5 * (ft[0:3] + ft[3:6])
        quat = quat_z2vec(vec / length)
        size[0] = size_in[0]
        size[1] = 0.5 * length
    else:
        size[:len(size_in)] = size_in[:3]
    return _Geom(name=elem.get("name", ""), type=gtype, size=size, pos=pos, quat=quat,
                 density=float(a["density"]), friction=_floats(a["friction"]), margin=float(a["margin"]),
                 gap=float(a["gap"]), condim=int(a["condim"]), contype=int(a["contype"]),
                 conaffinity=int(a["conaffinity"]), solref=_floats(a["solref"]), solimp=_floats(a["solimp"]),
                 solmix=float(a["solmix"]), priority=int(a["priority"]))


def _make_joint(elem: ET.Element, defaults: _Defaults, angle_scale: float) -> _Joint:
    a = defaults.resolve("joint", _JOINT_BUILTIN, dict(elem.attrib))
    _check_known("joint", a, list(_JOINT_BUILTIN))
    if float(a["frictionloss"]) != 0.0 or float(a["stiffness"]) != 0.0:
        raise MjcfError("joint frictionloss/stiffness not supported")
    if a["type"] == "free":
        jtype = JNT_FREE
    elif a["type"] == "hinge":
        jtype = JNT_HINGE
    else:
        raise MjcfError("unsupported joint type %r" % a["type"])
    axis = _floats(a["axis"])
    n = np.linalg.norm(axis)
    if n < MJ_MINVAL:
        raise MjcfError("zero joint axis")
    rng = _floats(a["range"])
    limited = _bool(a["limited"])
    if jtype == JNT_HINGE:
        rng = rng * angle_scale
        ref = float(a["ref"]) * angle_scale
    else:
        limited = False
        ref = 0.0
    return _Joint(name=elem.get("name", ""), type=jtype, pos=_floats(a["pos"]), axis=axis / n, limited=limited,
                  range=rng, armature=float(a["armature"]), damping=float(a["damping"]),
                  stiffness=float(a["stiffness"]), margin=float(a["margin"]), ref=ref)


def _make_body(elem: ET.Element, defaults: _Defaults, angle_scale: float) -> _Body:
    _check_known("body", elem.attrib, ["pos", "quat"])
    body = _Body(name=elem.get("name", ""), pos=_floats(elem.get("pos", "0 0 0")),
                 quat=_floats(elem.get("quat", "1 0 0 0")))
    body.quat = body.quat / np.linalg.norm(body.quat)
    for child in elem:
        if child.tag == "geom":
            body.geoms.append(_make_geom(child, defaults, angle_scale))
        elif child.tag == "joint":
            body.joints.append(_make_joint(child, defaults, angle_scale))
        elif child.tag == "body":
            body.children.append(_make_body(child, defaults, angle_scale))
        elif child.tag in ("light", "camera", "site"):
            continue
        else:
            raise MjcfError("unsupported element <%s> inside <body>" % child.tag)
    return body


def _prefix_names(elem: ET.Element, attr: str, prefix: str):
    for e in elem.iter():
        v = e.get(attr)
        if v is not None:
            e.set(attr, prefix + "/" + v)


# ----------------------------------------------------------------------------------------------
# compiled model
# ----------------------------------------------------------------------------------------------
_INT_FIELDS = [
    "body_parentid", "body_rootid", "body_weldid", "body_jntnum", "body_jntadr", "body_dofnum", "body_dofadr",
    "body_geomadr", "jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited",
    "dof_bodyid", "dof_jntid", "dof_parentid", "geom_type", "geom_bodyid", "geom_condim",
    "actuator_dofid", "pair_geom1", "pair_geom2",
    "agent_qposadr", "agent_nq", "agent_dofadr", "agent_nv", "agent_bodyadr", "agent_nbody", "agent_uadr",
    "agent_nu", "agent_torso",
]
_FLT_FIELDS = [
    "opt",  # timestep, gx, gy, gz, tolerance, meaninertia, impratio, iterations
    "qpos0", "body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia",
    "body_subtreemass", "body_invweight0", "jnt_pos", "jnt_axis", "jnt_range", "jnt_margin",
    "dof_armature", "dof_damping", "dof_invweight0", "geom_size", "geom_pos", "geom_quat", "geom_rbound",
    "geom_friction", "geom_margin", "geom_gap", "geom_solref", "geom_solimp", "geom_solmix",
    "actuator_gear", "actuator_ctrlrange",
    "pair_margin", "pair_gap", "pair_friction", "pair_solref", "pair_solimp",
]
BLOB_MAGIC = 0x4F4D5553  # 'SUMO'
BLOB_VERSION = 3


@dataclass
class SumoModel:
    """Flat constant tables of one compiled two-agent scene (see module docstring)."""
    name: str
    nq: int
    nv: int
    nu: int
    nbody: int
    njnt: int
    ngeom: int
    npair: int
    nagent: int
    body_names: List[str]
    joint_names: List[str]
    geom_names: List[str]
    tables: Dict[str, np.ndarray]
    # game constants (sumo.py:34-55)
    tatami_size: float = 2.0
    timestep_limit: int = 500
    frame_skip: int = 5

    def __getattr__(self, key):
        tables = self.__dict__.get("tables")
        if tables is not None and key in tables:
            return tables[key]
        raise AttributeError(key)

    # ---- agent views (agents.py:45-115) -------------------------------------------------------
    @property
    def obs_dims(self) -> List[int]:
        """``agents.py:190-214``: own qpos + own qvel + 6*own bodies + 7 + 6 + 1."""
        return [int(self.agent_nq[i] + self.agent_nv[i] + 6 * self.agent_nbody[i] + 7 + 6 + 1)
                for i in range(self.nagent)]

    @property
    def act_dims(self) -> List[int]:
        return [int(x) for x in self.agent_nu]

    def dims(self) -> Dict[str, int]:
        return dict(nq=self.nq, nv=self.nv, nu=self.nu, nbody=self.nbody, njnt=self.njnt, ngeom=self.ngeom,
                    npair=self.npair, nagent=self.nagent)

    # ---- serialisation -----------------------------------------------------------------------
    def to_blob(self) -> bytes:
        """Binary layout of ``include/sumo_model.h``: int32 header, then every int table, then
        (8-byte aligned) every float64 table, each in the fixed order of ``_INT_FIELDS`` /
        ``_FLT_FIELDS``; table lengths are implied by the header dims."""
        hdr = np.array([BLOB_MAGIC, BLOB_VERSION, self.nq, self.nv, self.nu, self.nbody, self.njnt, self.ngeom,
                        self.npair, self.nagent, self.frame_skip, self.timestep_limit, 0, 0, 0, 0], dtype=np.int32)
        ints = [hdr]
        for k in _INT_FIELDS:
            ints.append(np.ascontiguousarray(self.tables[k], dtype=np.int32).ravel())
        ibuf = np.concatenate(ints)
        if ibuf.size % 2:
            ibuf = np.concatenate([ibuf, np.zeros(1, np.int32)])
        flts = [np.array([self.tatami_size], dtype=np.float64)]
        for k in _FLT_FIELDS:
            flts.append(np.ascontiguousarray(self.tables[k], dtype=np.float64).ravel())
        fbuf = np.concatenate(flts)
        nints = np.array([ibuf.size, fbuf.size], dtype=np.int32)
        # first 8 bytes: counts, so the reader can locate the float section
        return nints.tobytes() + ibuf.tobytes() + fbuf.tobytes()

    def to_json(self) -> str:
        d = dict(name=self.name, nq=self.nq, nv=self.nv, nu=self.nu, nbody=self.nbody, njnt=self.njnt,
                 ngeom=self.ngeom, npair=self.npair, nagent=self.nagent, body_names=self.body_names,
                 joint_names=self.joint_names, geom_names=self.geom_names, tatami_size=self.tatami_size,
                 timestep_limit=self.timestep_limit, frame_skip=self.frame_skip,
                 tables={k: dict(dtype=str(v.dtype), shape=list(v.shape), data=v.ravel().tolist())
                         for k, v in self.tables.items()})
        return json.dumps(d)

    @staticmethod
    def from_json(s: str) -> "SumoModel":
        d = json.loads(s)
        tables = {k: np.array(v["data"], dtype=v["dtype"]).reshape(v["shape"]) for k, v in d.pop("tables").items()}
        return SumoModel(tables=tables, **d)


# ----------------------------------------------------------------------------------------------
# geometry -> inertia
# ----------------------------------------------------------------------------------------------
def _geom_mass_inertia(g: _Geom) -> Tuple[float, np.ndarray]:
    """Mass and principal inertia (in the geom frame) of a uniform-density primitive."""
    rho = g.density
    if g.type == GEOM_SPHERE:
        r = g.size[0]
        m = rho * 4.0 / 3.0 * math.pi * r ** 3
        return m, np.full(3, 0.4 * m * r * r)
    if g.type == GEOM_CAPSULE:
        r, h = g.size[0], 2.0 * g.size[1]
        mc = rho * math.pi * r * r * h
        ms = rho * 4.0 / 3.0 * math.pi * r ** 3
        ixy = mc * (3 * r * r + h * h) / 12.0 + 0.4 * ms * r * r + ms * h * (3 * r + 2 * h) / 8.0
        iz = mc * r * r / 2.0 + 0.4 * ms * r * r
        return mc + ms, np.array([ixy, ixy, iz])
    if g.type == GEOM_CYLINDER:
        r, h = g.size[0], 2.0 * g.size[1]
        m = rho * math.pi * r * r * h
        ixy = m * (3 * r * r + h * h) / 12.0
        return m, np.array([ixy, ixy, m * r * r / 2.0])
    if g.type == GEOM_BOX:
        a, b, c = g.size
        m = rho * 8 * a * b * c
        return m, m / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    raise MjcfError("cannot derive inertia from geom type %d" % g.type)


def _geom_rbound(g: _Geom) -> float:
    if g.type == GEOM_SPHERE:
        return g.size[0]
    if g.type == GEOM_CAPSULE:
        return g.size[0] + g.size[1]
    if g.type == GEOM_CYLINDER:
        return math.sqrt(g.size[0] ** 2 + g.size[1] ** 2)
    if g.type == GEOM_BOX:
        return float(np.linalg.norm(g.size))
    return 0.0  # plane


# ----------------------------------------------------------------------------------------------
# the compiler proper
# ----------------------------------------------------------------------------------------------
def compile_scene(scene_xml: str, agent_xmls: Sequence[str], agent_names: Sequence[str],
                  agent_densities: Optional[Sequence[float]] = None, tatami_size: Optional[float] = 2.0,
                  timestep_limit: int = 500, frame_skip: int = 5, init_poses=None,
                  name: str = "sumo") -> SumoModel:
    """Assemble (``utils.py:46-183``) and compile the two-agent scene.

    ``scene_xml`` / ``agent_xmls`` are paths to MJCF files laid out like the reference's
    ``assets/tatami.xml`` and ``assets/{ant,bug,spider}.xml``.
    """
    n_agents = len(agent_xmls)
    if n_agents != 2:
        raise MjcfError("Only 2-agent sumo is supported (utils.py:53)")
    scopes = ["%s%d" % (nm, i) for i, nm in enumerate(agent_names)]          # sumo.py:65-68
    if agent_densities is None:
        agent_densities = [10.0] * n_agents                                  # utils.py:97-98

    root = ET.parse(scene_xml).getroot()
    comp = root.find("compiler")
    comp_attr = dict(comp.attrib) if comp is not None else {}
    angle_scale = math.pi / 180.0 if comp_attr.get("angle", "degree") == "degree" else 1.0
    if comp_attr.get("coordinate", "local") != "local":
        raise MjcfError("only local coordinates supported")
    if comp_attr.get("inertiafromgeom", "auto") not in ("true", "auto"):
        raise MjcfError("inertiafromgeom must be true/auto")
    opt = root.find("option")
    opt_attr = dict(opt.attrib) if opt is not None else {}
    for k in opt_attr:
        if k not in ("integrator", "timestep"):
            raise MjcfError("unsupported <option %s>" % k)
    if opt_attr.get("integrator", "Euler") != "RK4":
        raise MjcfError("engine implements the RK4 integrator only (tatami.xml:3)")
    timestep = float(opt_attr.get("timestep", "0.002"))

    defaults = _Defaults()
    defaults.load_main(root.find("default"))

    # --- world body -------------------------------------------------------------------------------
    wb = root.find("worldbody")
    if tatami_size is not None:                                             # utils.py:64-88
        s = float("%.2f" % tatami_size)
        t = float("%.2f" % (tatami_size + 0.3))
        border = {"topborder": (-s, s, s, s), "rightborder": (s, -s, s, s),
                  "bottomborder": (-s, -s, s, -s), "leftborder": (-s, -s, -s, s)}
        for g in wb.findall("geom"):
            nm = g.get("name")
            if nm == "tatami":
                g.set("size", "%r %r 0.25" % (t, t))
            elif nm in border:
                x0, y0, x1, y1 = border[nm]
                g.set("fromto", "%r %r 0.5 %r %r 0.5" % (x0, y0, x1, y1))
    world = _make_body(wb, defaults, angle_scale)
    world.name = "world"
    if world.joints or world.children:
        raise MjcfError("world file must hold static geoms only")

    if init_poses is None:                                                  # utils.py:107-115
        r, phi, z = 1.5, 0.0, 0.75
        delta = (2.0 * np.pi) / n_agents
        init_poses = [(r * np.cos(phi + i * delta), r * np.sin(phi + i * delta), z) for i in range(n_agents)]

    motors: List[_Motor] = []
    agent_roots: List[_Body] = []
    for i in range(n_agents):
        aroot = ET.parse(agent_xmls[i]).getroot()
        adef = aroot.find("default")
        defaults.add_class(scopes[i], list(adef) if adef is not None else [],
                           {"density": repr(float(agent_densities[i]))})
        abody = aroot.find("body")
        abody.set("pos", " ".join(repr(float(v)) for v in init_poses[i]))
        for g in abody.iter("geom"):                                        # utils.py:151
            g.set("class", scopes[i])
        _prefix_names(abody, "name", scopes[i])                             # utils.py:153
        agent_roots.append(_make_body(abody, defaults, angle_scale))
        act = aroot.find("actuator")
        for m in (list(act) if act is not None else []):
            if m.tag != "motor":
                raise MjcfError("unsupported actuator <%s>" % m.tag)
            attr = dict(m.attrib)
            attr["class"] = scopes[i]                                        # utils.py:162
            a = defaults.resolve("motor", _MOTOR_BUILTIN, attr)
            _check_known("motor", a, list(_MOTOR_BUILTIN) + ["joint"])
            motors.append(_Motor(name=scopes[i] + "/" + m.get("name", "motor%d" % len(motors)),
                                 joint=scopes[i] + "/" + a["joint"], gear=_floats(a["gear"])[0],
                                 ctrllimited=_bool(a["ctrllimited"]), ctrlrange=_floats(a["ctrlrange"])))

    return _compile(world, agent_roots, motors, scopes, timestep, name, tatami_size if tatami_size is not None
                    else 2.0, timestep_limit, frame_skip)


def _compile(world: _Body, agent_roots: List[_Body], motors: List[_Motor], scopes: List[str], timestep: float,
             name: str, tatami_size: float, timestep_limit: int, frame_skip: int) -> SumoModel:
    # ---- flatten bodies depth-first (MuJoCo body order) --------------------------------------------
    bodies: List[_Body] = []
    parent: List[int] = []

    def visit(b: _Body, pid: int):
        bid = len(bodies)
        bodies.append(b)
        parent.append(pid)
        for c in b.children:
            visit(c, bid)

    visit(world, 0)
    agent_bodyadr, agent_nbody = [], []
    for ar in agent_roots:
        agent_bodyadr.append(len(bodies))
        visit(ar, 0)
        agent_nbody.append(len(bodies) - agent_bodyadr[-1])
    nbody = len(bodies)

    T: Dict[str, np.ndarray] = {}
    body_parentid = np.array(parent, dtype=np.int32)
    body_rootid = np.zeros(nbody, np.int32)
    body_weldid = np.zeros(nbody, np.int32)
    body_jntnum = np.zeros(nbody, np.int32)
    body_jntadr = np.full(nbody, -1, np.int32)
    body_dofnum = np.zeros(nbody, np.int32)
    body_dofadr = np.full(nbody, -1, np.int32)
    body_geomadr = np.full(nbody + 1, 0, np.int32)

    jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited = [], [], [], [], []
    jnt_pos, jnt_axis, jnt_range, jnt_margin, qpos0 = [], [], [], [], []
    dof_bodyid, dof_jntid, dof_parentid, dof_armature, dof_damping = [], [], [], [], []
    joint_names, geom_names = [], []
    geoms: List[_Geom] = []
    geom_bodyid: List[int] = []
    last_dof_of_body = np.full(nbody, -1, np.int32)

    nq = nv = 0
    for b, body in enumerate(bodies):
        pid = parent[b]
        if b == 0:
            body_rootid[b] = 0
        else:
            body_rootid[b] = b if pid == 0 else body_rootid[pid]
        body_weldid[b] = b if (body.joints or b == 0) else body_weldid[pid]
        if any(j.type == JNT_FREE for j in body.joints) and (len(body.joints) != 1 or pid != 0):
            raise MjcfError("free joint must be alone in a top-level body")
        body_jntnum[b] = len(body.joints)
        if body.joints:
            body_jntadr[b] = len(jnt_type)
            body_dofadr[b] = nv
        # parent dof for the first dof of this body = last dof of the closest ancestor that has one
        anc = pid
        pdof = -1
        if b != 0:
            pdof = last_dof_of_body[anc]
        for j in body.joints:
            jid = len(jnt_type)
            joint_names.append(j.name)
            jnt_type.append(j.type)
            jnt_qposadr.append(nq)
            jnt_dofadr.append(nv)
            jnt_bodyid.append(b)
            jnt_limited.append(int(j.limited))
            jnt_pos.append(j.pos)
            jnt_axis.append(j.axis)
            jnt_range.append(j.range)
            jnt_margin.append(j.margin)
            if j.type == JNT_FREE:
                # qpos0 of a free joint = body frame in the world (top-level body)
                qpos0.extend(list(body.pos) + list(body.quat))
                nq += 7
                ndof = 6
            else:
                qpos0.append(j.ref)
                nq += 1
                ndof = 1
            for _ in range(ndof):
                dof_bodyid.append(b)
                dof_jntid.append(jid)
                dof_parentid.append(pdof)
                dof_armature.append(j.armature)
                dof_damping.append(j.damping)
                pdof = nv
                nv += 1
        body_dofnum[b] = (nv - body_dofadr[b]) if body.joints else 0
        last_dof_of_body[b] = pdof
        body_geomadr[b] = len(geoms)
        for g in body.geoms:
            geoms.append(g)
            geom_bodyid.append(b)
            geom_names.append(g.name)
    body_geomadr[nbody] = len(geoms)
    njnt, ngeom = len(jnt_type), len(geoms)
    qpos0 = np.array(qpos0, dtype=np.float64)

    # ---- body frames + inertia from geoms ---------------------------------------------------------
    body_pos = np.array([b.pos for b in bodies])
    body_quat = np.array([b.quat for b in bodies])
    body_mass = np.zeros(nbody)
    body_inertia = np.zeros((nbody, 3))
    body_ipos = np.zeros((nbody, 3))
    body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (nbody, 1))
    for b, body in enumerate(bodies):
        if b == 0 or not body.geoms:
            if b != 0:
                raise MjcfError("body %s has no geoms: mass undefined" % body.name)
            continue
        if len(body.geoms) == 1:
            g = body.geoms[0]
            m, inert = _geom_mass_inertia(g)
            body_mass[b], body_inertia[b], body_ipos[b], body_iquat[b] = m, inert, g.pos, g.quat
        else:
            ms, coms, its = [], [], []
            for g in body.geoms:
                m, inert = _geom_mass_inertia(g)
                R = quat2mat(g.quat)
                ms.append(m)
                coms.append(g.pos)
                its.append(R @ np.diag(inert) @ R.T)
            mt = sum(ms)
            com = sum(m * c for m, c in zip(ms, coms)) / mt
            I = np.zeros((3, 3))
            for m, c, it in zip(ms, coms, its):
                d = c - com
                I += it + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            w, V = np.linalg.eigh(I)
            if np.linalg.det(V) < 0:
                V[:, 2] = -V[:, 2]
            # rotation matrix -> quaternion
            q = _mat2quat(V)
            body_mass[b], body_inertia[b], body_ipos[b], body_iquat[b] = mt, w, com, q
    subtree = body_mass.copy()
    for b in range(nbody - 1, 0, -1):
        subtree[parent[b]] += subtree[b]

    T.update(body_parentid=body_parentid, body_rootid=body_rootid, body_weldid=body_weldid,
             body_jntnum=body_jntnum, body_jntadr=body_jntadr, body_dofnum=body_dofnum, body_dofadr=body_dofadr,
             body_geomadr=body_geomadr,
             jnt_type=np.array(jnt_type, np.int32), jnt_qposadr=np.array(jnt_qposadr, np.int32),
             jnt_dofadr=np.array(jnt_dofadr, np.int32), jnt_bodyid=np.array(jnt_bodyid, np.int32),
             jnt_limited=np.array(jnt_limited, np.int32),
             dof_bodyid=np.array(dof_bodyid, np.int32), dof_jntid=np.array(dof_jntid, np.int32),
             dof_parentid=np.array(dof_parentid, np.int32),
             qpos0=qpos0, body_pos=body_pos, body_quat=body_quat, body_ipos=body_ipos, body_iquat=body_iquat,
             body_mass=body_mass, body_inertia=body_inertia, body_subtreemass=subtree,
             jnt_pos=np.array(jnt_pos), jnt_axis=np.array(jnt_axis), jnt_range=np.array(jnt_range),
             jnt_margin=np.array(jnt_margin), dof_armature=np.array(dof_armature),
             dof_damping=np.array(dof_damping))

    # ---- geoms ---------------------------------------------------------------------------------
    T.update(geom_type=np.array([g.type for g in geoms], np.int32), geom_bodyid=np.array(geom_bodyid, np.int32),
             geom_condim=np.array([g.condim for g in geoms], np.int32),
             geom_size=np.array([g.size for g in geoms]), geom_pos=np.array([g.pos for g in geoms]),
             geom_quat=np.array([g.quat for g in geoms]), geom_rbound=np.array([_geom_rbound(g) for g in geoms]),
             geom_friction=np.array([g.friction for g in geoms]), geom_margin=np.array([g.margin for g in geoms]),
             geom_gap=np.array([g.gap for g in geoms]), geom_solref=np.array([g.solref for g in geoms]),
             geom_solimp=np.array([g.solimp for g in geoms]), geom_solmix=np.array([g.solmix for g in geoms]))

    # ---- actuators -------------------------------------------------------------------------------
    act_dof, act_gear, act_range = [], [], []
    for m in motors:
        if m.joint not in joint_names:
            raise MjcfError("motor joint %r not found" % m.joint)
        jid = joint_names.index(m.joint)
        if jnt_type[jid] != JNT_HINGE:
            raise MjcfError("motors must drive hinge joints")
        if not m.ctrllimited:
            raise MjcfError("unlimited motors not supported")
        act_dof.append(jnt_dofadr[jid])
        act_gear.append(m.gear)
        act_range.append(m.ctrlrange)
    nu = len(motors)
    T.update(actuator_dofid=np.array(act_dof, np.int32), actuator_gear=np.array(act_gear),
             actuator_ctrlrange=np.array(act_range))

    # ---- static collision pair list (mj_collision body-pair order + filterBodyPair) ----------------
    weldparent = np.array([body_weldid[parent[body_weldid[b]]] for b in range(nbody)], np.int32)
    p1, p2, pm, pg, pf, psr, psi = [], [], [], [], [], [], []
    for b1 in range(nbody):
        for b2 in range(b1 + 1, nbody):
            w1, w2 = body_weldid[b1], body_weldid[b2]
            if w1 == w2:
                continue
            if w1 != 0 and w2 != 0 and (w1 == weldparent[b2] or w2 == weldparent[b1]):
                continue
            for g1 in range(body_geomadr[b1], body_geomadr[b1 + 1]):
                for g2 in range(body_geomadr[b2], body_geomadr[b2 + 1]):
                    A, B = geoms[g1], geoms[g2]
                    if not ((A.contype & B.conaffinity) or (B.contype & A.conaffinity)):
                        continue
                    a, c = (g1, g2) if A.type <= B.type else (g2, g1)       # lower geom type first
                    GA, GB = geoms[a], geoms[c]
                    if GA.priority != GB.priority:
                        raise MjcfError("geom priorities not supported")
                    if max(GA.condim, GB.condim) != 3:
                        raise MjcfError("only condim 3 contacts supported")
                    mix = GA.solmix / (GA.solmix + GB.solmix)
                    p1.append(a)
                    p2.append(c)
                    pm.append(max(GA.margin, GB.margin))
                    pg.append(max(GA.gap, GB.gap))
                    pf.append(np.maximum(GA.friction, GB.friction))
                    psr.append(mix * GA.solref + (1 - mix) * GB.solref)
                    psi.append(mix * GA.solimp + (1 - mix) * GB.solimp)
    npair = len(p1)
    T.update(pair_geom1=np.array(p1, np.int32), pair_geom2=np.array(p2, np.int32), pair_margin=np.array(pm),
             pair_gap=np.array(pg), pair_friction=np.array(pf), pair_solref=np.array(psr),
             pair_solimp=np.array(psi))

    # ---- agent index ranges (agents.py:45-83) ---------------------------------------------------
    a_qadr, a_nq, a_dadr, a_nv, a_uadr, a_nu, a_torso = [], [], [], [], [], [], []
    for i, scope in enumerate(scopes):
        jids = [k for k, nm in enumerate(joint_names) if nm.startswith(scope)]
        a_qadr.append(jnt_qposadr[jids[0]])
        a_nq.append(sum(7 if jnt_type[k] == JNT_FREE else 1 for k in jids))
        a_dadr.append(jnt_dofadr[jids[0]])
        a_nv.append(sum(6 if jnt_type[k] == JNT_FREE else 1 for k in jids))
        uids = [k for k, m in enumerate(motors) if m.name.startswith(scope)]
        a_uadr.append(uids[0])
        a_nu.append(len(uids))
        a_torso.append(agent_bodyadr[i])
    T.update(agent_qposadr=np.array(a_qadr, np.int32), agent_nq=np.array(a_nq, np.int32),
             agent_dofadr=np.array(a_dadr, np.int32), agent_nv=np.array(a_nv, np.int32),
             agent_bodyadr=np.array(agent_bodyadr, np.int32), agent_nbody=np.array(agent_nbody, np.int32),
             agent_uadr=np.array(a_uadr, np.int32), agent_nu=np.array(a_nu, np.int32),
             agent_torso=np.array(a_torso, np.int32))

    model = SumoModel(name=name, nq=nq, nv=nv, nu=nu, nbody=nbody, njnt=njnt, ngeom=ngeom, npair=npair,
                      nagent=len(scopes), body_names=[b.name for b in bodies], joint_names=joint_names,
                      geom_names=geom_names, tables=T, tatami_size=float(tatami_size),
                      timestep_limit=int(timestep_limit), frame_skip=int(frame_skip))
    _set_const(model, timestep)
    return model


def _mat2quat(R):
    tr = np.trace(R)
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


# ----------------------------------------------------------------------------------------------
# mj_setConst equivalent: constants that need the mass matrix at qpos0
# ----------------------------------------------------------------------------------------------
def kinematics_np(model: SumoModel, qpos: np.ndarray):
    """Body frames at ``qpos`` (numpy; used at compile time and by tests)."""
    nb = model.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    xanchor = np.zeros((model.njnt, 3))
    xaxis = np.zeros((model.njnt, 3))
    for b in range(1, nb):
        pid = model.body_parentid[b]
        ja, jn = model.body_jntadr[b], model.body_jntnum[b]
        if jn == 1 and model.jnt_type[ja] == JNT_FREE:
            qa = model.jnt_qposadr[ja]
            xpos[b] = qpos[qa:qa + 3]
            q = qpos[qa + 3:qa + 7]
            xquat[b] = q / np.linalg.norm(q)
            xanchor[ja] = xpos[b]
            xaxis[ja] = quat2mat(xquat[b]) @ model.jnt_axis[ja]
            continue
        Rp = quat2mat(xquat[pid])
        xpos[b] = xpos[pid] + Rp @ model.body_pos[b]
        xquat[b] = quat_mul(xquat[pid], model.body_quat[b])
        for j in range(ja, ja + jn):
            R = quat2mat(xquat[b])
            xanchor[j] = xpos[b] + R @ model.jnt_pos[j]
            xaxis[j] = R @ model.jnt_axis[j]
            ang = qpos[model.jnt_qposadr[j]] - model.qpos0[model.jnt_qposadr[j]]
            ql = np.concatenate([[math.cos(ang / 2)], model.jnt_axis[j] * math.sin(ang / 2)])
            xquat[b] = quat_mul(xquat[b], ql)
            xpos[b] = xanchor[j] - quat2mat(xquat[b]) @ model.jnt_pos[j]
        xquat[b] /= np.linalg.norm(xquat[b])
    return xpos, xquat, xanchor, xaxis


def mass_matrix_np(model: SumoModel, qpos: np.ndarray):
    """Joint-space inertia via body Jacobians: M = sum_b m Jp^T Jp + Jr^T I Jr + diag(armature).
    Deliberately a different formulation from the engine's composite-rigid-body pass, so tests can
    cross-check the two.  Returns (M, jacp_com[nbody,3,nv], jacr[nbody,3,nv])."""
    nb, nv = model.nbody, model.nv
    xpos, xquat, xanchor, xaxis = kinematics_np(model, qpos)
    xipos = np.zeros((nb, 3))
    for b in range(nb):
        xipos[b] = xpos[b] + quat2mat(xquat[b]) @ model.body_ipos[b]
    jacp = np.zeros((nb, 3, nv))
    jacr = np.zeros((nb, 3, nv))
    for b in range(1, nb):
        # walk the dof chain of b
        bb = b
        while bb != 0 and model.body_dofnum[bb] == 0:
            bb = model.body_parentid[bb]
        d = model.body_dofadr[bb] + model.body_dofnum[bb] - 1 if bb != 0 else -1
        while d >= 0:
            j = model.dof_jntid[d]
            if model.jnt_type[j] == JNT_FREE:
                k = d - model.jnt_dofadr[j]
                if k < 3:
                    jacp[b, k, d] = 1.0
                else:
                    ax = quat2mat(xquat[model.jnt_bodyid[j]])[:, k - 3]
                    jacr[b, :, d] = ax
                    jacp[b, :, d] = np.cross(ax, xipos[b] - xanchor[j])
            else:
                jacr[b, :, d] = xaxis[j]
                jacp[b, :, d] = np.cross(xaxis[j], xipos[b] - xanchor[j])
            d = model.dof_parentid[d]
    M = np.diag(model.dof_armature.astype(np.float64))
    for b in range(1, nb):
        R = quat2mat(quat_mul(xquat[b], model.body_iquat[b]))
        Iw = R @ np.diag(model.body_inertia[b]) @ R.T
        M += model.body_mass[b] * jacp[b].T @ jacp[b] + jacr[b].T @ Iw @ jacr[b]
    return M, jacp, jacr


def _set_const(model: SumoModel, timestep: float):
    """``mj_setConst`` for the quantities the constraint model needs."""
    nv, nb = model.nv, model.nbody
    M, jacp, jacr = mass_matrix_np(model, model.qpos0)
    Minv = np.linalg.inv(M)
    dof_inv = np.diag(Minv).copy()
    for j in range(model.njnt):
        if model.jnt_type[j] == JNT_FREE:
            d = model.jnt_dofadr[j]
            dof_inv[d:d + 3] = dof_inv[d:d + 3].mean()
            dof_inv[d + 3:d + 6] = dof_inv[d + 3:d + 6].mean()
    body_inv = np.zeros((nb, 2))
    for b in range(1, nb):
        J = np.vstack([jacp[b], jacr[b]])
        A = J @ Minv @ J.T
        body_inv[b, 0] = max(MJ_MINVAL, np.trace(A[:3, :3]) / 3.0)
        body_inv[b, 1] = max(MJ_MINVAL, np.trace(A[3:, 3:]) / 3.0)
    meaninertia = float(np.trace(M) / nv)
    model.tables["dof_invweight0"] = dof_inv
    model.tables["body_invweight0"] = body_inv
    # opt: timestep, gravity xyz, tolerance, meaninertia, impratio, iterations
    model.tables["opt"] = np.array([timestep, 0.0, 0.0, -9.81, 1e-8, meaninertia, 1.0, 100.0])


# ----------------------------------------------------------------------------------------------
# registry (robosumo/__init__.py:8-105) + packaged compiled scenes
# ----------------------------------------------------------------------------------------------
_DENSITY = {"ant": 13.0, "bug": 10.0, "spider": 39.0}
_ASSET_DIR = os.path.join(os.path.dirname(__file__), "assets")


def registry() -> Dict[str, Dict]:
    reg = {}
    names = ["ant", "bug", "spider"]
    for a in names:
        for b in names:
            env_id = "RoboSumo-%s-vs-%s-v0" % (a.capitalize(), b.capitalize())
            reg[env_id] = dict(agent_names=[a, b], agent_densities=[_DENSITY[a], _DENSITY[b]], tatami_size=2.0,
                               timestep_limit=500)
    # spellings used by BASELINE.json (not registered by the reference)
    reg["RoboSumoAnts-v0"] = reg["RoboSumo-Ant-vs-Ant-v0"]
    reg["RoboSumoSpiders-v0"] = reg["RoboSumo-Spider-vs-Spider-v0"]
    return reg


def canonical_id(env_id: str) -> str:
    alias = {"RoboSumoAnts-v0": "RoboSumo-Ant-vs-Ant-v0", "RoboSumoSpiders-v0": "RoboSumo-Spider-vs-Spider-v0"}
    return alias.get(env_id, env_id)


def compile_env(env_id: str, asset_dir: str) -> SumoModel:
    """Compile ``env_id`` from MJCF files in ``asset_dir`` (a directory laid out like the
    reference's ``robosumo/robosumo/envs/assets``)."""
    kw = registry()[env_id]
    return compile_scene(os.path.join(asset_dir, "tatami.xml"),
                         [os.path.join(asset_dir, n + ".xml") for n in kw["agent_names"]],
                         kw["agent_names"], kw["agent_densities"], kw["tatami_size"], kw["timestep_limit"],
                         name=canonical_id(env_id))


def load_model(env_id: str, asset_dir: Optional[str] = None) -> SumoModel:
    """Model for ``env_id``.  With ``asset_dir`` (or ``$ROBOSUMO_ASSETS``) the MJCF sources are compiled;
    otherwise the pre-compiled table shipped in ``robosumo_selfplay_amd/assets`` is loaded."""
    if env_id not in registry():
        raise KeyError("unknown env id %r" % env_id)
    asset_dir = asset_dir or os.environ.get("ROBOSUMO_ASSETS")
    if asset_dir:
        return compile_env(env_id, asset_dir)
    path = os.path.join(_ASSET_DIR, canonical_id(env_id) + ".json")
    if not os.path.exists(path):
        raise FileNotFoundError("no pre-compiled scene %s; set ROBOSUMO_ASSETS to an MJCF asset directory" % path)
    with open(path) as f:
        return SumoModel.from_json(f.read())
