"""Metashape camera-file parsing: the numeric inputs of the projection path.

Mirror of geograypher/utils/parsing.py:46-157 (`make_4x4_transform`, `parse_transform_metashape`, `parse_sensors`):
chunk -> EPSG:4978 4x4 transform, per-sensor `f, cx, cy, image_width, image_height, distortion_params`.
"""
import xml.etree.ElementTree as ET

import numpy as np


def make_4x4_transform(rotation_str: str, translation_str: str, scale_str: str = "1"):
    """4x4 homogeneous transform from Metashape's strings (reference: parsing.py:46-70).

    Raises:
        ValueError: when the 9 rotation entries are not a proper rotation (determinant 1).
    """
    rotation_np = np.array(rotation_str.split(), dtype=float).reshape(3, 3)
    if not np.isclose(np.linalg.det(rotation_np), 1.0, atol=1e-8, rtol=0):
        raise ValueError(f"Inproper rotation matrix with determinant {np.linalg.det(rotation_np)}")
    transform = np.eye(4)
    transform[:3, :3] = rotation_np * float(scale_str)
    transform[:3, 3] = np.array(translation_str.split(), dtype=float)
    return transform


def parse_transform_metashape(camera_file, return_component_id: bool = False):
    """Chunk -> EPSG:4978 transform of the ACTIVE component (reference: parsing.py:73-111)."""
    root = ET.parse(camera_file).getroot()
    components = root.find("chunk").find("components")
    active_component_id = components.get("active_id")
    active_component = components.find(f"component[@id='{active_component_id}']")
    transform = active_component.find("transform")
    if transform is None:
        local_to_epsg_4978_transform = None
    else:
        local_to_epsg_4978_transform = make_4x4_transform(
            transform.find("rotation").text, transform.find("translation").text, transform.find("scale").text
        )
    if return_component_id:
        return local_to_epsg_4978_transform, active_component_id
    return local_to_epsg_4978_transform


def parse_sensors(sensors, default_sensor_dict=None):
    """{sensor id: intrinsics dict or None} (reference: parsing.py:114-157)."""
    sensors_dict = {}
    for sensor in sensors:
        sensor_dict = {"image_width": int(sensor[0].get("width")), "image_height": int(sensor[0].get("height"))}
        calibration = sensor.find("calibration[@class='adjusted']")
        if calibration is None:
            if default_sensor_dict is not None:
                sensor_dict.update(default_sensor_dict)
            else:
                sensor_dict = None
        else:
            sensor_dict["f"] = float(calibration.find("f").text)
            cx, cy = calibration.find("cx"), calibration.find("cy")
            try:
                sensor_dict["cx"] = float(cx.text) if cx is not None else default_sensor_dict["cx"]
                sensor_dict["cy"] = float(cy.text) if cy is not None else default_sensor_dict["cy"]
                sensor_dict["distortion_params"] = {
                    child.tag: float(child.text)
                    for child in calibration
                    if child.tag not in ["resolution", "f", "cx", "cy"]
                }
            except (KeyError, TypeError):
                sensor_dict = None
        sensors_dict[int(sensor.get("id"))] = sensor_dict
    return sensors_dict
