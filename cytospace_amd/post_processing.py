"""Output writers of the reference: `save_results`
(/root/reference/cytospace/post_processing/post_processing.py:10-116) and the unassigned-spot table written by
main_cytospace (/root/reference/cytospace/cytospace.py:686-694).  Pure host code (pandas), same file names, columns and
ordering, so a run of this package can be post-processed by the same downstream scripts.  Written from the format, not
from the reference's statements: tests/golden/gv12_* holds files the reference itself wrote for the same inputs.
"""
import math
import os

import numpy as np
import pandas as pd


def _strip(prefix, names):
    n = len(prefix)
    return [str(x)[n:] for x in names]          # the reference drops the first five characters ('CELL_', 'SPOT_', ...)


def assigned_locations_table(cell_ids_selected, assigned_locations, cell_type_data, sampling_method):
    """The rows of <prefix>assigned_locations.csv: one per assigned cell, in assignment order.
    UniqueCID = 'UCID' + zero-padded running number (width = digits of the cell count); a cell that is not in
    cell_type_data is a place-holder: OriginalCID 'NA', its type read from its id 'CELL_<type>_new_<k>'."""
    ids = [str(c) for c in cell_ids_selected]
    width = int(math.log10(len(ids))) + 1
    known = set(cell_type_data.index)
    type_col = cell_type_data.columns[0]
    cols = {"UniqueCID": ["UCID" + str(i).zfill(width) for i in range(len(ids))],
            "OriginalCID": _strip("CELL_", [c if c in known else "CELL_NA" for c in ids])}
    if sampling_method == "place_holders":
        cols["PlaceHolderCID"] = _strip("CELL_", ids)
    cols["CellType"] = _strip("TYPE_", [cell_type_data.at[c, type_col] if c in known else "TYPE_" + c.split("_")[1] for c in ids])
    cols["SpotID"] = _strip("SPOT_", assigned_locations.index)
    for k in range(2):
        cols[assigned_locations.columns.values[k]] = list(assigned_locations.iloc[:, k])
    return pd.DataFrame.from_dict(cols)


def save_results(output_path, output_prefix, cell_ids_selected, all_cells_save, assigned_locations,
                 cell_type_data, sampling_method, single_cell):
    """post_processing.py:10-116, same arguments.  Writes
      <prefix>assigned_locations.csv
      <prefix>assigned_expression/{genes.tsv, barcodes.tsv, matrix.mtx}     (sampling_method "duplicates")
      <prefix>new_scRNA.csv                                                 ("place_holders": the generated cells)
      <prefix>cell_type_assignments_by_spot.csv, <prefix>fractional_abundances_by_spot.csv   (not in single-cell mode)"""
    import scipy.io
    import scipy.sparse
    df = assigned_locations_table(cell_ids_selected, assigned_locations, cell_type_data, sampling_method)
    df.to_csv(os.path.join(output_path, f"{output_prefix}assigned_locations.csv"), index=False)

    if sampling_method == "duplicates":
        out_dir = os.path.join(output_path, f"{output_prefix}assigned_expression")
        if os.path.exists(out_dir):
            print("\033[91mWARNING\033[0m: {} exists and the expression matrix may be overwritten.".format(out_dir))
        os.makedirs(out_dir, exist_ok=True)
        expr = all_cells_save.loc[:, ["CELL_{}".format(x) for x in df.OriginalCID]]
        expr.index = _strip("GENE_", expr.index)
        expr.columns = df.UniqueCID
        genes = expr.index.to_frame()
        genes.reset_index(inplace=True)           # gene id twice: Read10X expects the name in the second column
        genes.to_csv(os.path.join(out_dir, "genes.tsv"), sep="\t", header=False, index=False)
        expr.columns.to_frame().to_csv(os.path.join(out_dir, "barcodes.tsv"), sep="\t", header=False, index=False)
        scipy.io.mmwrite(os.path.join(out_dir, "matrix.mtx"), scipy.sparse.coo_matrix(expr))
    else:
        generated = all_cells_save.loc[:, ~all_cells_save.columns.isin(cell_type_data.index)]
        generated.index = _strip("GENE_", generated.index.astype(str))
        generated.columns = _strip("CELL_", generated.columns.astype(str))
        generated.to_csv(os.path.join(output_path, f"{output_prefix}new_scRNA.csv"))

    if not single_cell:
        # SpotID x CellType counts, spots and types in order of first appearance, plus the row total
        counts = df.loc[:, ["SpotID", "CellType"]].value_counts().unstack(fill_value=0) \
            .reindex(index=df.SpotID.unique(), columns=df.CellType.unique())
        counts["Total cells"] = counts.sum(axis=1)
        counts = counts.astype(int)
        counts.index.name = "SpotID"
        counts.to_csv(os.path.join(output_path, f"{output_prefix}cell_type_assignments_by_spot.csv"))
        totals = np.array(counts["Total cells"], dtype=float)
        counts.iloc[:, :-1].div(totals, axis=0).to_csv(os.path.join(output_path, f"{output_prefix}fractional_abundances_by_spot.csv"))


def save_unassigned_locations(output_path, output_prefix, all_spot_ids, assigned_locations, coordinates_data):
    """cytospace.py:686-694: the spots that received no cell, with a 'Number of cells' column of zeros.
    Returns the number of such spots (the file is only written when there are any)."""
    unmapped = np.setdiff1d(list(all_spot_ids), list(assigned_locations.index)).tolist()
    if unmapped:
        table = coordinates_data.loc[unmapped].copy()
        table.index = table.index.str.replace("SPOT_", "")
        table["Number of cells"] = 0
        table.to_csv(f"{output_path}/{output_prefix}unassigned_locations.csv", index=True)
    return len(unmapped)
