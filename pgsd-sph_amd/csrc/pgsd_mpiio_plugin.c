/* pgsd_mpiio_plugin.c -- libpgsd_amd_mpiio.so: the MPI-IO back end of libpgsd_amd.so (pgsd_mpiio_plugin.h).
 *
 * The reference ends its write path in MPI_File_write_at (pgsd.c:2229, 1154, 2032) and reads with MPI_File_read_at
 * (pgsd.c:2534); with PGSD_IO=mpiio so does this library, at the identical offsets.  Independent (non-collective)
 * calls, like the reference's.  The file is opened on MPI_COMM_SELF: the reference's collective MPI_File_open on
 * MPI_COMM_WORLD (pgsd.c:1748) buys nothing for independent I/O, and the library's communicator need not be an MPI
 * communicator at all.  Compiled with the MPI it runs under; needs MPI_THREAD_SERIALIZED when the device pipeline's
 * writer thread is in play (the library serialises its calls into this plugin). */
#include "pgsd_mpiio_plugin.h"

#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[MPI_MAX_ERROR_STRING + 64];

static int fail(const char* what, int rc)
    {
    char msg[MPI_MAX_ERROR_STRING];
    int n = 0;
    msg[0] = 0;
    if (rc != MPI_SUCCESS)
        MPI_Error_string(rc, msg, &n);
    snprintf(g_err, sizeof(g_err), "%s: %s", what, msg);
    return -1;
    }

static int mpiio_open(const char* path, int readonly, void** fh)
    {
    MPI_File* f = (MPI_File*)malloc(sizeof(MPI_File));
    if (!f)
        return fail("out of memory", MPI_SUCCESS);
    int rc = MPI_File_open(MPI_COMM_SELF, (char*)path, readonly ? MPI_MODE_RDONLY : MPI_MODE_RDWR, MPI_INFO_NULL, f);
    if (rc != MPI_SUCCESS)
        {
        free(f);
        return fail("MPI_File_open", rc);
        }
    *fh = f;
    return 0;
    }

static int mpiio_close(void* fh)
    {
    MPI_File* f = (MPI_File*)fh;
    int rc = MPI_File_close(f);
    free(f);
    return rc == MPI_SUCCESS ? 0 : fail("MPI_File_close", rc);
    }

#define PIECE (1ll << 30)

static long long mpiio_write_at(void* fh, long long offset, const void* buf, long long bytes)
    {
    MPI_File f = *(MPI_File*)fh;
    long long done = 0;
    while (done < bytes)
        {
        const int n = (int)(bytes - done < PIECE ? bytes - done : PIECE);
        MPI_Status st;
        int rc = MPI_File_write_at(f, (MPI_Offset)(offset + done), (char*)buf + done, n, MPI_BYTE, &st);
        if (rc != MPI_SUCCESS)
            return fail("MPI_File_write_at", rc);
        int got = 0;
        MPI_Get_count(&st, MPI_BYTE, &got);
        if (got <= 0)
            return fail("MPI_File_write_at wrote nothing", MPI_SUCCESS);
        done += got;
        }
    return done;
    }

static long long mpiio_read_at(void* fh, long long offset, void* buf, long long bytes)
    {
    MPI_File f = *(MPI_File*)fh;
    long long done = 0;
    while (done < bytes)
        {
        const int n = (int)(bytes - done < PIECE ? bytes - done : PIECE);
        MPI_Status st;
        int rc = MPI_File_read_at(f, (MPI_Offset)(offset + done), (char*)buf + done, n, MPI_BYTE, &st);
        if (rc != MPI_SUCCESS)
            return fail("MPI_File_read_at", rc);
        int got = 0;
        MPI_Get_count(&st, MPI_BYTE, &got);
        if (got <= 0)
            break; /* end of file */
        done += got;
        }
    return done;
    }

static int mpiio_set_size(void* fh, long long size)
    {
    int rc = MPI_File_set_size(*(MPI_File*)fh, (MPI_Offset)size);
    return rc == MPI_SUCCESS ? 0 : fail("MPI_File_set_size", rc);
    }

static long long mpiio_get_size(void* fh)
    {
    MPI_Offset size = 0;
    int rc = MPI_File_get_size(*(MPI_File*)fh, &size);
    return rc == MPI_SUCCESS ? (long long)size : fail("MPI_File_get_size", rc);
    }

static const char* mpiio_last_error(void)
    {
    return g_err;
    }

int pgsd_mpiio_plugin(struct pgsd_mpiio_api* out)
    {
    int on = 0;
    if (MPI_Initialized(&on) != MPI_SUCCESS || !on)
        {
        snprintf(g_err, sizeof(g_err), "MPI is not initialised (the caller initialises MPI, as with the reference)");
        if (out)
            out->last_error = mpiio_last_error;
        return -1;
        }
    out->open = mpiio_open;
    out->close = mpiio_close;
    out->write_at = mpiio_write_at;
    out->read_at = mpiio_read_at;
    out->set_size = mpiio_set_size;
    out->get_size = mpiio_get_size;
    out->last_error = mpiio_last_error;
    return 0;
    }
