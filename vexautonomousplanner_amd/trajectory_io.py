"""Trajectory and route file formats of the reference — host-side formatting
only, no numerics: the values come from generate_motion_profile (device).

  trajectory rows   gui/gui_manager.py:284-315   [0, t, x*12, -y*12, heading, v*12, omega] per time step (inches,
                                                 y negated), with a [1, *action_values] row inserted in front of
                                                 the step where each node / action point is reached
  .txt              gui/gui_manager.py:220-230   one row per line, values separated (and followed) by one space
  routes.h entry    gui/gui_manager.py:442-507   std::vector<std::vector<double>> <name> = {{..}, {..}};
  route JSON        gui/gui_manager.py:388-427   [[node rows], [action-point rows]], compact separators
"""
import json
from typing import List, Sequence


def trajectory_rows(profile, node_action_values: Sequence[Sequence[float]],
                    action_point_values: Sequence[Sequence[float]] = ()) -> List[list]:
    """profile: the 9-tuple of generate_motion_profile with nodes_map already extended by len(times)
    (gui/path.py:342), i.e. one entry per node."""
    times, _positions, vels, _accs, headings, omegas, nodes_map, actions_map, coords = profile
    rows = [[0, times[i], coords[i][0] * 12, coords[i][1] * -12, headings[i], vels[i] * 12, omegas[i]]
            for i in range(len(times))]
    for i in range(len(nodes_map)):
        rows.insert(int(nodes_map[i]) + i, [1] + list(node_action_values[i]))
    for i in range(len(actions_map)):
        rows.insert(int(actions_map[i]) + i, [1] + list(action_point_values[i]))
    return rows


def format_txt(rows) -> str:
    return "".join("".join(f"{v} " for v in row) + "\n" for row in rows)


def write_trajectory_txt(path: str, rows) -> None:
    with open(path, "w") as f:
        f.write(format_txt(rows))


def routes_header_entry(name: str, rows) -> str:
    parts = []
    for row in rows:
        if len(row) > 2:
            parts.append("{" + ", ".join(f"{v}" for v in row) + "}")
        else:
            parts.append(f"{{{row[0]}, {row[1]}}}")
    return f"std::vector<std::vector<double>> {name} = {{{', '.join(parts)}}};\n"


def update_routes_header(path: str, name: str, rows) -> None:
    """Replace the route's line in routes.h, insert it before #endif, or create the file."""
    entry = routes_header_entry(name, rows)
    try:
        with open(path) as f:
            content = f.readlines()
    except FileNotFoundError:
        content = ["#ifndef ROUTES_H\n", "#define ROUTES_H\n", "#include <vector>\n", "\n", entry, "\n", "#endif\n"]
    else:
        prefix = f"std::vector<std::vector<double>> {name} ="
        for i, line in enumerate(content):
            if line.strip().startswith(prefix):
                content[i] = entry
                break
        else:
            for i, line in enumerate(content):
                if line.strip() == "#endif":
                    content.insert(i, entry)
                    break
            else:
                content.append(entry)
    with open(path, "w") as f:
        f.writelines(content)


def route_json(nodes_data, action_data) -> str:
    return json.dumps([nodes_data, action_data], separators=(",", ":"))


def parse_route_json(text: str):
    nodes_data, action_data = json.loads(text)
    return nodes_data, action_data
